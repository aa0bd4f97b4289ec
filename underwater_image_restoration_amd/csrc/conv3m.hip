// uir_conv3x3_mfma: dense 3x3 convolutions (pad 1, no bias) with at most 32 channels on either side - the network's ends and its
// first two samplers with their input gradients: outputl 32 -> 8 and its transpose 8 -> 32 (SpectralTransformer.py:247,269), down1
// 16 -> 8 / 8 -> 16 (:154), down2 32 -> 16 / 16 -> 32 (:157).
//
// These are contractions over K = 9 x channels (72 ... 288) with 8 ... 32 outputs per pixel on 0.26 ... 1 M pixels: 150 ... 170 MB per
// launch at full resolution, i.e. ~35 us of HBM time.  The panel kernel of csrc/gemm.hip gathers every tap's A rows from global memory
// (rows of 32 ... 128 bytes, nine times) and ran them at 0.9 ... 1.1 TB/s; the VALU kernel of csrc/conv3s.hip pays 72 scalar weight loads
// per input channel and reaches a third of the VALU rate (180 us for outputl).  Here:
//   * a persistent workgroup walks 16 x 16 pixel tiles; the 18 x 18 x K input tile (with ring) goes through LDS once, and the NEXT
//     tile's rows are already in flight (registers) while the current one is contracted;
//   * v_mfma_f32_16x16x4_f32 with the WEIGHTS as the A operand (rows = output channels, held in registers for the whole launch) and
//     the pixels as the B operand (16 consecutive pixels of an image row = the 16 columns): D[n][pixel], so a lane ends with four
//     consecutive output channels of one pixel - float4 stores, 16-byte residual loads; 8 outputs use lane groups 0 and 1 only
//     (an MFMA row tile is 16 wide: half of it idle there - the kernel is bound by its bytes, not by the matrix pipe);
//   * the contraction index of a k-step is any bijection both operands share: lane group j takes channels 4 j .. 4 j + 3 of a
//     16-channel block (one ds_read_b128 feeds four MFMAs).
// Reached through uir_gemm (taps = 9, plain epilogue, no bias): no ABI of its own.  Same tap / flip semantics as csrc/conv3s.hip.
#include "uir_common.h"
#include "../../include/uir_kernels.h"

namespace {

typedef float f32x4m __attribute__((ext_vector_type(4)));

template <int KC>
struct Conv3m {
  static constexpr int KQ = KC / 4;
  static constexpr int KS = (KQ & 1) ? KC : KC + 4;        // LDS pixel stride (floats); KS / 4 odd: conflict-free ds_read_b128 / b64
  static constexpr int CPL = KC >= 16 ? 4 : 2;             // channels per lane and channel block
  static constexpr int NCB = KC / (4 * CPL);               // channel blocks per tap (32: 2, 16: 1, 8: 1)
  static constexpr int TW = 16, TH = 16, RW = TW + 2, RH = TH + 2;
  static constexpr int NE = RH * RW * KQ;                  // float4 slots of the input tile
  static constexpr int NLD = (NE + 255) / 256;
  static constexpr int LDS_BYTES = RH * RW * KS * 4;
};

template <int KC, int NT, bool FLIP, bool HAS_R>
__global__ __launch_bounds__(256, 2) void uir_conv3x3_mfma_kernel(const UirGemmArgs a, const int ntiles) {
  using T = Conv3m<KC>;
  extern __shared__ __attribute__((aligned(16))) float tile[];       // [RH * RW][KS]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l16 = lane & 15, j16 = lane >> 4;
  const int tiles_x = (a.Wd + T::TW - 1) / T::TW, tiles_y = (a.H + T::TH - 1) / T::TH;

  // ---- weights -> A-operand registers, once per workgroup: wv[nt][t][cb][s] = W[n = 16 nt + l16][channel cb * 4 CPL + CPL j16 + s][tap t] ----
  float wv[NT][9][T::NCB][T::CPL];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int n = 16 * nt + l16;
    const int nr = n < a.N ? uir_colmap(n, a.n_grp_real, a.n_grp_pad) : -1;
#pragma unroll
    for (int cb = 0; cb < T::NCB; ++cb)
#pragma unroll
      for (int s = 0; s < T::CPL; ++s) {
        const int k = cb * 4 * T::CPL + T::CPL * j16 + s;
        const int kr = k < a.K ? uir_colmap(k, a.k_grp_real, a.k_grp_pad) : -1;
        const bool ok = nr >= 0 && kr >= 0;
        const float* wp = a.W + (ok ? (long)nr * a.w_sn + (long)kr * a.w_sk : 0);
#pragma unroll
        for (int t = 0; t < 9; ++t) {
          const float w = wp[FLIP ? 8 - t : t];
          wv[nt][t][cb][s] = ok ? w : 0.f;
        }
      }
  }

  struct Tc { int y0, x0; long img; };
  auto coords = [&](int tl) {
    int bid = tl < ntiles ? tl : ntiles - 1;                // past the end: a valid tile (its prefetch is never consumed)
    const int tx = bid % tiles_x;
    bid /= tiles_x;
    const int ty = bid % tiles_y;
    Tc c;
    c.y0 = ty * T::TH; c.x0 = tx * T::TW; c.img = (long)(bid / tiles_y) * a.H * a.Wd;
    return c;
  };
  // input tile + ring: raw values from clamped addresses (prefetch), zeroed outside the image (Conv2d's zero padding) at the LDS store
  float4 stage[T::NLD];
  auto load_tile = [&](const Tc& c) {
#pragma unroll
    for (int t = 0; t < T::NLD; ++t) {
      const int e = min(tid + 256 * t, T::NE - 1);
      const int p = e / T::KQ, q = e - p * T::KQ;
      const int ry = p / T::RW, rx = p - ry * T::RW;
      const int iy = min(max(c.y0 + ry - 1, 0), a.H - 1), ix = min(max(c.x0 + rx - 1, 0), a.Wd - 1);
      stage[t] = *reinterpret_cast<const float4*>(a.X + (c.img + (long)iy * a.Wd + ix) * a.ldx + 4 * q);
    }
  };
  load_tile(coords(blockIdx.x));

  for (int tl = blockIdx.x; tl < ntiles; tl += gridDim.x) {
    const Tc c = coords(tl);
    __syncthreads();                                        // every wave is past its reads of the previous tile
#pragma unroll
    for (int t = 0; t < T::NLD; ++t) {
      const int e = tid + 256 * t;
      const int p = e / T::KQ, q = e - p * T::KQ;
      const int ry = p / T::RW, rx = p - ry * T::RW;
      const int iy = c.y0 + ry - 1, ix = c.x0 + rx - 1;
      const float z = (iy >= 0 && iy < a.H && ix >= 0 && ix < a.Wd) ? 1.f : 0.f;
      if (e < T::NE)
        *reinterpret_cast<float4*>(tile + p * T::KS + 4 * q) = make_float4(stage[t].x * z, stage[t].y * z, stage[t].z * z, stage[t].w * z);
    }
    __syncthreads();
    load_tile(coords(tl + gridDim.x));                       // lands under the MFMAs below

    // this lane's pixel of image row 4 wave + r of the tile; residual rows first (used last)
    const int ix = c.x0 + l16;
    float4 rv[HAS_R ? 4 : 1][HAS_R ? NT : 1];
    if constexpr (HAS_R) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int iy = c.y0 + 4 * wave + r;
        const long m = c.img + (long)min(iy, a.H - 1) * a.Wd + min(ix, a.Wd - 1);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          const int n = 16 * nt + 4 * j16;
          rv[r][nt] = n + 4 <= a.N ? *reinterpret_cast<const float4*>(a.R + m * a.ldr + n) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
      }
    }
    f32x4m acc[4][NT];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) acc[r][nt] = f32x4m{0.f, 0.f, 0.f, 0.f};
    const float* bp0 = tile + ((4 * wave) * T::RW + l16) * T::KS + T::CPL * j16;
    // blocks = (tap, channel block); block i + 1's four LDS reads are issued before block i's MFMAs (pinned: hoisted freely, all 18
    // blocks' reads took 288 VGPRs; sunk next to their MFMAs, every block exposed an LDS round trip)
    constexpr int NBLK = 9 * T::NCB;
    float bv[2][4][T::CPL];
    auto fetch = [&](const int blk, float (&o)[4][T::CPL]) {
      const int t = blk / T::NCB, cb = blk - t * T::NCB;
      const int dy = t / 3, dx = t - 3 * dy;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float* bp = bp0 + ((r + dy) * T::RW + dx) * T::KS + cb * 4 * T::CPL;
        if constexpr (T::CPL == 4) {
          const float4 v = *reinterpret_cast<const float4*>(bp);
          o[r][0] = v.x; o[r][1] = v.y; o[r][2] = v.z; o[r][3] = v.w;
        } else {
          const float2 v = *reinterpret_cast<const float2*>(bp);
          o[r][0] = v.x; o[r][1] = v.y;
        }
      }
    };
    fetch(0, bv[0]);
#pragma unroll
    for (int blk = 0; blk < NBLK; ++blk) {
      const int t = blk / T::NCB, cb = blk - t * T::NCB;
      if (blk + 1 < NBLK) fetch(blk + 1, bv[(blk + 1) & 1]);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int s = 0; s < T::CPL; ++s)
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt)
            acc[r][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[nt][t][cb][s], bv[blk & 1][r][s], acc[r][nt], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    // D[n = 4 j16 + i][pixel l16]: four consecutive output channels of one pixel per lane
    if (ix < a.Wd) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int iy = c.y0 + 4 * wave + r;
        if (iy < a.H) {
          const long m = c.img + (long)iy * a.Wd + ix;
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) {
            const int n = 16 * nt + 4 * j16;
            float4 v = make_float4(acc[r][nt][0], acc[r][nt][1], acc[r][nt][2], acc[r][nt][3]);
            if constexpr (HAS_R) { v.x += rv[r][nt].x; v.y += rv[r][nt].y; v.z += rv[r][nt].z; v.w += rv[r][nt].w; }
            if (n + 4 <= a.N) *reinterpret_cast<float4*>(a.Y + m * a.ldy + n) = v;
          }
        }
      }
    }
  }
}

template <int KC, int NT, bool FLIP>
int conv3m_launch(const UirGemmArgs& a, hipStream_t st) {
  using T = Conv3m<KC>;
  const int tiles = (a.M / (a.H * a.Wd)) * ((a.H + 15) / 16) * ((a.Wd + 15) / 16);
  // all workgroups resident (3 per CU by LDS at K = 32), the same number of tiles each
  const int per = (tiles + 767) / 768;
  const int grid = (tiles + per - 1) / per;
  if (a.R) hipLaunchKernelGGL((uir_conv3x3_mfma_kernel<KC, NT, FLIP, true>), dim3(grid), dim3(256), (size_t)T::LDS_BYTES, st, a, tiles);
  else hipLaunchKernelGGL((uir_conv3x3_mfma_kernel<KC, NT, FLIP, false>), dim3(grid), dim3(256), (size_t)T::LDS_BYTES, st, a, tiles);
  return uir_launch_status();
}

// ---- the weight gradient of the same convolutions: dW[n][k][tap] = sum_px dY[px][n] * X[px + off(tap)][k] (uir_gram, taps = 9) ----
// N <= 16 outputs, K <= 32 inputs (outputl, output, down1, down2): the nine-tap column kernel of csrc/gram.hip gathers nine shifted copies of
// dY per pixel pair from global memory and ran these four launches at 0.03-0.04 of the HBM peak (0.27 ms each on the bulk lane).  Here the
// X tile with its ring and the dY tile go through LDS once per 16 x 16 pixel tile; v_mfma_f32_16x16x4_f32 contracts over PIXELS:
// A = dY^T (row = output channel, k-slot j16 = pixel x + 4 j16 of an image row), B = X shifted by the tap (column = input channel of a
// 16-channel block), one 16 x 16 accumulator per (tap, channel block) held across all the workgroup's tiles - 18 of them for outputl.
// The two pixels a 32-lane half reads sit 4 apart: 4 * 36 floats = 16 banks.  Rows / columns past N / K multiply padding and are never stored.
template <int KC, int NQ>
__global__ __launch_bounds__(256, 2) void uir_conv3x3_wgrad_kernel(const UirGramArgs a, const int ntiles) {
  using T = Conv3m<KC>;
  constexpr int NCB = KC >= 32 ? 2 : 1;
  constexpr int NS = 20;                                    // dY tile row stride (floats): 4 * 20 = 16 banks between the two pixels of a half
  constexpr int NEY = 256 * NQ, NLY = (NEY + 255) / 256;
  extern __shared__ __attribute__((aligned(16))) float tile[];       // X: [RH * RW][KS] | dY: [256][NS]
  float* ys = tile + T::RH * T::RW * T::KS;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l16 = lane & 15, j16 = lane >> 4;
  const int tiles_x = (a.Wd + 15) / 16, tiles_y = (a.H + 15) / 16;
  struct Tc { int y0, x0; long img; };
  auto coords = [&](int tl) {
    int bid = tl < ntiles ? tl : ntiles - 1;
    const int tx = bid % tiles_x;
    bid /= tiles_x;
    const int ty = bid % tiles_y;
    Tc c;
    c.y0 = ty * 16; c.x0 = tx * 16; c.img = (long)(bid / tiles_y) * a.H * a.Wd;
    return c;
  };
  float4 sx[T::NLD], sy[NLY];
  auto load_tile = [&](const Tc& c) {
#pragma unroll
    for (int t = 0; t < T::NLD; ++t) {
      const int e = min(tid + 256 * t, T::NE - 1);
      const int p = e / T::KQ, q = e - p * T::KQ;
      const int ry = p / T::RW, rx = p - ry * T::RW;
      const int iy = min(max(c.y0 + ry - 1, 0), a.H - 1), ix = min(max(c.x0 + rx - 1, 0), a.Wd - 1);
      sx[t] = *reinterpret_cast<const float4*>(a.B + (c.img + (long)iy * a.Wd + ix) * a.ldb + 4 * q);
    }
#pragma unroll
    for (int t = 0; t < NLY; ++t) {
      const int e = tid + 256 * t, p = e / NQ, q = e - p * NQ;
      const int iy = min(c.y0 + (p >> 4), a.H - 1), ix = min(c.x0 + (p & 15), a.Wd - 1);
      sy[t] = *reinterpret_cast<const float4*>(a.A + (c.img + (long)iy * a.Wd + ix) * a.lda + 4 * q);
    }
  };
  f32x4m acc[9][NCB];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) acc[t][cb] = f32x4m{0.f, 0.f, 0.f, 0.f};
  load_tile(coords(blockIdx.x));

  for (int tl = blockIdx.x; tl < ntiles; tl += gridDim.x) {
    const Tc c = coords(tl);
    __syncthreads();
#pragma unroll
    for (int t = 0; t < T::NLD; ++t) {
      const int e = tid + 256 * t;
      const int p = e / T::KQ, q = e - p * T::KQ;
      const int ry = p / T::RW, rx = p - ry * T::RW;
      const int iy = c.y0 + ry - 1, ix = c.x0 + rx - 1;
      const float z = (iy >= 0 && iy < a.H && ix >= 0 && ix < a.Wd) ? 1.f : 0.f;
      if (e < T::NE) *reinterpret_cast<float4*>(tile + p * T::KS + 4 * q) = make_float4(sx[t].x * z, sx[t].y * z, sx[t].z * z, sx[t].w * z);
    }
#pragma unroll
    for (int t = 0; t < NLY; ++t) {
      const int e = tid + 256 * t, p = e / NQ, q = e - p * NQ;
      const float z = (c.y0 + (p >> 4) < a.H && c.x0 + (p & 15) < a.Wd) ? 1.f : 0.f;     // pixels outside the image add nothing
      if (e < NEY) *reinterpret_cast<float4*>(ys + p * NS + 4 * q) = make_float4(sy[t].x * z, sy[t].y * z, sy[t].z * z, sy[t].w * z);
    }
    __syncthreads();
    load_tile(coords(tl + gridDim.x));

    // this wave: image rows 4 wave .. 4 wave + 3 of the tile, 4 k-steps (16 pixels) per row; step (r, q): pixel x = q + 4 j16
    const float* ap0 = ys + ((4 * wave) * 16 + 4 * j16) * NS + l16;
    const float* bp0 = tile + ((4 * wave) * T::RW + 4 * j16) * T::KS + l16;
    float av[2], bv[2][9][NCB];
    auto fetch = [&](const int st, float& ao, float (&bo)[9][NCB]) {
      const int r = st >> 2, q = st & 3;
      ao = ap0[(r * 16 + q) * NS];
#pragma unroll
      for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) bo[t][cb] = bp0[((r + t / 3) * T::RW + q + t % 3) * T::KS + 16 * cb];
    };
    // operands of step st + 1 are read while step st's MFMAs issue; the widest variant (K = 32, N = 16) has no registers for the second
    // set (it spilled 60 bytes): its reads precede their own MFMAs
    constexpr bool PIPE = !(KC == 32 && NQ == 4);
    if (PIPE) fetch(0, av[0], bv[0]);
#pragma unroll
    for (int st = 0; st < 16; ++st) {
      const int cur = PIPE ? (st & 1) : 0;
      if (PIPE) {
        if (st + 1 < 16) fetch(st + 1, av[(st + 1) & 1], bv[(st + 1) & 1]);
      } else {
        fetch(st, av[0], bv[0]);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb)
          acc[t][cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[cur], bv[cur][t][cb], acc[t][cb], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  // the four waves' partials meet in LDS in a fixed order; one slab row per workgroup
  __syncthreads();
  float* red = tile;
  for (int w = 0; w < 4; ++w) {
    if (wave == w) {
#pragma unroll
      for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            float* p = &red[((t * NCB + cb) * 4 + i) * 64 + lane];
            *p = (w == 0) ? acc[t][cb][i] : (*p + acc[t][cb][i]);
          }
    }
    __syncthreads();
  }
  float* slab = a.slab + (long)blockIdx.x * a.slab_stride;
  for (int idx = tid; idx < 9 * NCB * 256; idx += 256) {
    const int ln = idx & 63, i = (idx >> 6) & 3, tc = idx >> 8;
    const int t = tc / NCB, cb = tc - t * NCB;
    const int n = 4 * (ln >> 4) + i, k = 16 * cb + (ln & 15);
    if (n < a.N && k < a.K) {
      const int nr = uir_colmap(n, a.n_grp_real, a.n_grp_pad);
      const int kr = uir_colmap(k, a.k_grp_real, a.k_grp_pad);
      if (nr >= 0 && kr >= 0) slab[(long)nr * a.o_sn + (long)kr * a.o_sk + (long)t * a.o_st] = red[idx];
    }
  }
}

template <int KC, int NQ>
int conv3m_wgrad_launch(const UirGramArgs& a, hipStream_t st) {
  using T = Conv3m<KC>;
  const int tiles = (a.M / (a.H * a.Wd)) * ((a.H + 15) / 16) * ((a.Wd + 15) / 16);
  const size_t lds = (size_t)T::LDS_BYTES + 256 * 20 * sizeof(float);
  hipLaunchKernelGGL((uir_conv3x3_wgrad_kernel<KC, NQ>), dim3(a.nchunks), dim3(256), lds, st, a, tiles);
  return uir_launch_status();
}

}  // namespace

// Weight gradient of a 3 x 3 convolution with N <= 16 outputs and K <= 32 inputs; -1: not a shape this kernel serves (uir_gram continues).
// One slab row per workgroup, a.nchunks workgroups (any count >= 1: workgroup w takes tiles w, w + nchunks, ...).
int uir_conv3x3_wgrad(const UirGramArgs& a, void* stream) {
  if (a.taps != 9 || a.batches != 1 || a.pro_b != 0 || a.sq_a_off >= 0 || a.sq_b_off >= 0 || a.cs_a_off >= 0) return -1;
  if (a.N > 16 || a.N < 4 || (a.N & 3) || a.K > 32 || a.K < 8 || (a.K & 3) || (a.lda & 3) || (a.ldb & 3) || !uir_aligned16(a.A) ||
      !uir_aligned16(a.B))
    return -1;
  const int Kc = a.K <= 8 ? 8 : a.K <= 16 ? 16 : 32;
  if (a.ldb < Kc || a.K != Kc) return -1;          // X rows are read in float4 groups up to the template width
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int nq = a.N / 4;
#define UIR_C3W(KV, QV) if (Kc == KV && nq == QV) return conv3m_wgrad_launch<KV, QV>(a, st)
  UIR_C3W(32, 2); UIR_C3W(32, 4); UIR_C3W(16, 2); UIR_C3W(8, 1);
#undef UIR_C3W
  return -1;
}

// -1: not a shape this kernel serves (the caller continues with csrc/conv3s.hip / the MFMA panel paths)
int uir_conv3x3_mfma(const UirGemmArgs& a, void* stream) {
  if (a.taps != 9 || a.ntap > 0 || a.batches != 1 || a.pro != UIR_PRO_NONE || a.epi != UIR_EPI_PLAIN || a.act != UIR_ACT_NONE || a.bias ||
      a.out_scale || a.pre_out)
    return -1;
  if (a.w_st != 1 || a.K > 32 || a.K < 8 || a.N > 32 || a.N < 8 || (a.N & 3) || (a.ldx & 3) || (a.ldy & 3) || !uir_aligned16(a.X) ||
      !uir_aligned16(a.Y))
    return -1;
  if (a.R && ((a.ldr & 3) || !uir_aligned16(a.R))) return -1;
  const int Kc = a.K <= 8 ? 8 : a.K <= 16 ? 16 : 32;
  const int nt = a.N <= 16 ? 1 : 2;
  if (a.ldx < Kc) return -1;                      // rows are read in whole float4 groups up to the template width
  hipStream_t st = static_cast<hipStream_t>(stream);
#define UIR_C3M(KV, NV, FV) if (Kc == KV && nt == NV && (a.flip != 0) == FV) return conv3m_launch<KV, NV, FV>(a, st)
  UIR_C3M(32, 1, false); UIR_C3M(16, 1, false); UIR_C3M(8, 2, true); UIR_C3M(8, 1, true); UIR_C3M(16, 2, true);
#undef UIR_C3M
  return -1;
}
