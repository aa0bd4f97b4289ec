// uir_conv3x3_small: dense 3x3 convolutions (pad 1, no bias) with at most EIGHT channels on both sides - the network's last layer
// output 8 -> 3 (SpectralTransformer.py:252,270) and its input gradient 3 -> 8 (the same contraction with the weight's channel axes
// swapped and the taps mirrored).  (Rounds 2-3 also served outputl 32 -> 8 and down1 16 -> 8 here; since round 4 every shape with
// 8 .. 32 channels on both sides runs on the 16x16x4 MFMA kernel of csrc/conv3m.hip.)
// On the MFMA path of uir_gemm a 32-column tile is 3/4 ... 7/8 empty for 8 or 4 output channels and the nine-tap panel walk costs
// 9 x K/32 chunk trips per row tile: 220 us for outputl at 16 x 256 x 256, 0.7 TB/s.  Here: one thread per pixel, its <= 8 outputs in
// registers, the 18 x 18 x K input tile of a 16 x 16 pixel workgroup tile in LDS, the weights as SCALAR operands (uniform
// addresses: s_load_dwordx8, no LDS / VGPR traffic for them; read back through LDS they would cost the bandwidth of ordinary
// reads, 16 B x 64 lanes per broadcast), v_pk_fma_f32 over output pairs with the window value broadcast (op_sel_hi on src0: not
// the erratum form of DESIGN.md 17, and build.lint_isa checks every build).
// Reached through uir_gemm (taps = 9, plain epilogue, no bias): no ABI of its own.
#include "uir_common.h"
#include "../../include/uir_kernels.h"

namespace {

template <int K, int N, bool FLIP>
__global__ __launch_bounds__(256) void uir_conv3x3_small_kernel(const UirGemmArgs a) {
  constexpr int KQ = K / 4;
  constexpr int KS = (KQ & 1) ? K : K + 4;               // LDS pixel stride (floats): KS / 4 odd -> conflict-free ds_read_b128
  constexpr int TW = 16, TH = 16, RW = TW + 2, RH = TH + 2;
  extern __shared__ __attribute__((aligned(16))) float tile[];   // [RH * RW][KS]
  const int tiles_x = (a.Wd + TW - 1) / TW, tiles_y = (a.H + TH - 1) / TH;
  int bid = blockIdx.x;
  const int tx = bid % tiles_x;
  bid /= tiles_x;
  const int ty = bid % tiles_y;
  const int b = bid / tiles_y;
  const int y0 = ty * TH, x0 = tx * TW;
  const long img = (long)b * a.H * a.Wd;
  const float* __restrict__ X = a.X;
  const float* __restrict__ Wg = a.W;                     // restrict: the weights are not written here -> eligible for scalar loads
  float* __restrict__ Yg = a.Y;

  // input tile + ring -> LDS, zero outside the image (Conv2d's zero padding); k >= real K columns are read as stored (the caller pads
  // activations with exact zeros, and their weights are zero by the column map below)
  // all of a thread's loads are issued before the first LDS store (a load -> store chain per trip costs one HBM latency per trip,
  // ten trips for K = 32, with only three workgroups per CU to hide it behind)
  constexpr int NE = RH * RW * KQ, NT = (NE + 255) / 256;
  float4 stage[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int e = min((int)threadIdx.x + 256 * t, NE - 1);
    const int p = e / KQ, q = e - p * KQ;
    const int ry = p / RW, rx = p - ry * RW;
    const int iy = y0 + ry - 1, ix = x0 + rx - 1;
    const bool ok = iy >= 0 && iy < a.H && ix >= 0 && ix < a.Wd;
    const float4 v = *reinterpret_cast<const float4*>(X + (img + (long)(ok ? iy : 0) * a.Wd + (ok ? ix : 0)) * a.ldx + 4 * q);
    const float z = ok ? 1.f : 0.f;
    stage[t] = make_float4(v.x * z, v.y * z, v.z * z, v.w * z);
  }
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int e = (int)threadIdx.x + 256 * t;
    if (e < NE) *reinterpret_cast<float4*>(tile + (e / KQ) * KS + 4 * (e % KQ)) = stage[t];
  }
  __syncthreads();

  const int py = threadIdx.x >> 4, px = threadIdx.x & 15;
  const int iy = y0 + py, ix = x0 + px;
  const bool inside = iy < a.H && ix < a.Wd;
  const long m = img + (long)(inside ? iy : 0) * a.Wd + (inside ? ix : 0);
  float* yp = Yg + m * a.ldy;
  const float* rp = a.R ? a.R + m * a.ldr : nullptr;
  // The nine taps of one (output channel, input channel) pair are contiguous in the Conv2d weight (w_st == 1, forward and
  // transposed alike): one run of scalar loads (s_load_dwordx8 + s_load_dword) meets the nine window values of that input channel.
  // Eight output channels per pass, one input channel per trip: 72 weights live in SGPRs (all K x N x 9 at once spilled 350-1500
  // SGPRs; a compiler fence instead makes the loads ineligible for the scalar path: no "memory" clobbers in this kernel).
  const float* xc = tile + ((py + 1) * RW + (px + 1)) * KS;
  constexpr int NB = N < 8 ? N : 8;
#pragma unroll
  for (int n0 = 0; n0 < N; n0 += NB) {
    int wn[NB];                                             // weight offsets of the pass's output channels (uniform), -1 = zero pad column
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      const int n = n0 + j;
      const int nr = n < a.N ? uir_colmap(n, a.n_grp_real, a.n_grp_pad) : -1;
      wn[j] = nr < 0 ? -1 : nr * (int)a.w_sn;
    }
    float acc[NB];
#pragma unroll
    for (int j = 0; j < NB; ++j) acc[j] = 0.f;
#pragma unroll 1
    for (int k = 0; k < K; ++k) {
      const int kr = k < a.K ? uir_colmap(k, a.k_grp_real, a.k_grp_pad) : -1;
      const int ko = kr < 0 ? -1 : kr * (int)a.w_sk;
      float x9[9];
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) x9[tap] = xc[((tap / 3 - 1) * RW + (tap % 3 - 1)) * KS + k];
#pragma unroll
      for (int j = 0; j < NB; ++j) {
        const bool ok = ko >= 0 && wn[j] >= 0;
        const float* __restrict__ wp = Wg + (ok ? wn[j] + ko : 0);   // uniform address: scalar loads
        float s_ = 0.f;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) s_ = fmaf(x9[tap], wp[FLIP ? 8 - tap : tap], s_);
        acc[j] += ok ? s_ : 0.f;
      }
    }
    if (inside) {
#pragma unroll
      for (int j = 0; j < NB; j += 4) {
        const int n = n0 + j;
        float4 v = make_float4(acc[j], acc[j + 1], acc[j + 2], acc[j + 3]);
        if (rp) {
          const float4 r = *reinterpret_cast<const float4*>(rp + n);
          v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w;
        }
        if (n + 4 <= a.N) {
          *reinterpret_cast<float4*>(yp + n) = v;
        } else {
          const float vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (n + e < a.N) yp[n + e] = vv[e];
        }
      }
    }
  }
}

template <int K, int N, bool FLIP>
int conv3s_launch(const UirGemmArgs& a, hipStream_t st) {
  constexpr int KQ = K / 4, KS = (KQ & 1) ? K : K + 4;
  const int tiles = (a.M / (a.H * a.Wd)) * ((a.H + 15) / 16) * ((a.Wd + 15) / 16);
  hipLaunchKernelGGL((uir_conv3x3_small_kernel<K, N, FLIP>), dim3(tiles), dim3(256), (size_t)18 * 18 * KS * sizeof(float), st, a);
  return uir_launch_status();
}

}  // namespace

// -1: not a shape this kernel serves (the caller continues with the MFMA paths)
int uir_conv3x3_small(const UirGemmArgs& a, void* stream) {
  if (a.taps != 9 || a.ntap > 0 || a.batches != 1 || a.pro != UIR_PRO_NONE || a.epi != UIR_EPI_PLAIN || a.act != UIR_ACT_NONE || a.bias)
    return -1;
  if (a.w_st != 1 || a.K > 32 || a.N > 32 || (a.ldx & 3) || (a.ldy & 3) || !uir_aligned16(a.X) || !uir_aligned16(a.Y)) return -1;
  if (a.R && ((a.ldr & 3) || !uir_aligned16(a.R))) return -1;
  const int Kc = a.K <= 4 ? 4 : a.K <= 8 ? 8 : a.K <= 16 ? 16 : 32;
  const int Nc = a.N <= 4 ? 4 : a.N <= 8 ? 8 : a.N <= 16 ? 16 : 32;
  if (a.ldx < Kc) return -1;                     // rows are read in whole float4 groups up to the template width
  hipStream_t st = static_cast<hipStream_t>(stream);
#define UIR_C3(KV, NV, FV) if (Kc == KV && Nc == NV && (a.flip != 0) == FV) return conv3s_launch<KV, NV, FV>(a, st)
  // measured against the MFMA path at 16 x 256 x 256 (16 x 128 x 128 for down2): outputl 179 vs 220 us, output 32 vs 108, down1 89 vs 148,
  // output's input gradient 33 vs 109.  With 16 or 32 outputs (two or four passes, each waiting on its own scalar loads) the MFMA
  // path wins (outputl.dX 253 vs 111 us, down2 125 vs 68): those shapes are not served here.
  // (round 4: outputl 32 -> 8 and down1 16 -> 8 moved to csrc/conv3m.hip, which uir_gemm asks first: 110 vs 179 us, 54 vs 83 us)
  UIR_C3(8, 4, false); UIR_C3(4, 8, true);
#undef UIR_C3
  return -1;
}
