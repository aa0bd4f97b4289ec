// uir_gemm: the channel-contraction workhorse (1x1 / 3x3 dense convolutions on HWC rows,
// forward and input-gradient), on the f32 matrix cores of gfx950.
//
//   Y[m, n] = epi( sum_tap sum_k  pro(X[nbr(m,tap), k]) * W[n, k, tap]  + bias[n] ) (+ R[m, n])
//
// Rows m are HWC pixels (M = B*H*W, or H*W per batch z when weights are per-batch); columns are
// channels.  One wave owns 32 pixels x (32*NT) output channels: v_mfma_f32_32x32x2_f32 with the
// pixel on the MFMA row (accumulator register) and the channel on the lane, so every epilogue
// store is a 128-B contiguous channel run.  The A operand is read straight from HBM as one
// float4 per lane (k = 8j + 4*half .. +3, the k-order inside an 8-chunk is permuted, which a sum
// over k does not care about); the B operand (weights, <= a few 100 KB, L1/L2 resident) is read
// through the same k permutation with arbitrary (n, k, tap) strides, so the reference's native
// Conv2d layout [N][K][kh][kw] serves forward (sn=K*taps, sk=taps) and backward (swapped) alike.
//
// Prologues (applied to A on load):  LN  = per-row LayerNorm over the K channels (stats written
// out for backward), GATE = gelu(x[k]) * x[gate_off + k]  (GDFN gate, never materialised).
// Epilogues: bias, LeakyReLU, residual add, GATE_BWD (dgate -> dg1, dg2), LEAKY_BWD.
#include "uir_common.h"
#include "../../include/uir_kernels.h"

namespace {

template <int NT>
__global__ __launch_bounds__(256) void uir_gemm_kernel(const UirGemmArgs a) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 31, h2 = lane >> 5;
  const int z = blockIdx.z;
  const int m0 = blockIdx.x * 128 + wave * 32;
  if (m0 >= a.M) return;  // no block-level barrier below: whole waves may leave
  const int n0 = blockIdx.y * (32 * NT);
  const float* __restrict__ X = a.X + (long)z * a.x_bs;
  const float* __restrict__ W = a.W + (long)z * a.w_bs;
  const int m = m0 + r;
  const bool mvalid = m < a.M;
  int ph = 0, pw = 0;
  if (a.taps > 1) {
    int hw = m % (a.H * a.Wd);
    ph = hw / a.Wd;
    pw = hw - ph * a.Wd;
  }

  // ---- LayerNorm prologue: row statistics over the K channels (two-pass, like ATen) ----
  float mean = 0.f, rstd = 0.f;
  if (a.pro == UIR_PRO_LN) {
    const float* xr = X + (long)m * a.ldx;
    float s = 0.f;
    for (int kc = 4 * h2; kc < a.K; kc += 8) {
      if (mvalid) { float4 v = *reinterpret_cast<const float4*>(xr + kc); s += (v.x + v.y) + (v.z + v.w); }
    }
    s += __shfl_xor(s, 32);
    mean = s / (float)a.K;
    float q = 0.f;
    for (int kc = 4 * h2; kc < a.K; kc += 8) {
      if (mvalid) {
        float4 v = *reinterpret_cast<const float4*>(xr + kc);
        float d0 = v.x - mean, d1 = v.y - mean, d2 = v.z - mean, d3 = v.w - mean;
        q += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
      }
    }
    q += __shfl_xor(q, 32);
    rstd = rsqrtf(q / (float)a.K + a.ln_eps);
    if (a.stats && blockIdx.y == 0 && h2 == 0 && mvalid) {
      float* st = a.stats + ((long)z * a.M + m) * 2;
      st[0] = mean;
      st[1] = rstd;
    }
  }

  f32x16 acc[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[nt][i] = 0.f;

  const int ksteps = (a.K + 7) >> 3;
  const int nsteps = a.taps * ksteps;

  // real weight row of each of this lane's output columns (-1: exact-zero pad / out of range)
  long wrow[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    int n = n0 + 32 * nt + r;
    int nr = (n < a.N) ? uir_colmap(n, a.n_grp_real, a.n_grp_pad) : -1;
    wrow[nt] = nr < 0 ? -1 : (long)nr * a.w_sn;
  }

  int ld_tap = 0, ld_ks = 0;  // position of the next step to load
  auto load_step = [&](float (&av)[4], float (&bv)[NT][4]) {
    const int tap = ld_tap;
    const int kc = ld_ks * 8 + 4 * h2;
    bool ok = mvalid && kc < a.K;
    long row = m;
    if (a.taps > 1) {
      int dy = tap / 3 - 1, dx = tap - (tap / 3) * 3 - 1;
      int hh = ph + dy, ww = pw + dx;
      ok = ok && hh >= 0 && hh < a.H && ww >= 0 && ww < a.Wd;
      row = (long)m + dy * a.Wd + dx;
    }
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (ok) {
      const float* xp = X + row * a.ldx + kc;
      v = *reinterpret_cast<const float4*>(xp);
      if (a.pro == UIR_PRO_LN) {
        float4 g = *reinterpret_cast<const float4*>(a.ln_g + kc);
        float4 b = *reinterpret_cast<const float4*>(a.ln_b + kc);
        v.x = (v.x - mean) * rstd * g.x + b.x;
        v.y = (v.y - mean) * rstd * g.y + b.y;
        v.z = (v.z - mean) * rstd * g.z + b.z;
        v.w = (v.w - mean) * rstd * g.w + b.w;
      } else if (a.pro == UIR_PRO_GATE) {
        float4 u = *reinterpret_cast<const float4*>(xp + a.gate_off);
        v.x = uir_gelu(v.x) * u.x;
        v.y = uir_gelu(v.y) * u.y;
        v.z = uir_gelu(v.z) * u.z;
        v.w = uir_gelu(v.w) * u.w;
      }
    }
    av[0] = v.x; av[1] = v.y; av[2] = v.z; av[3] = v.w;
    const int wt = a.flip ? (a.taps - 1 - tap) : tap;
    long koff[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      int k = kc + i;
      int kr = (k < a.K) ? uir_colmap(k, a.k_grp_real, a.k_grp_pad) : -1;
      koff[i] = kr < 0 ? -1 : (long)kr * a.w_sk + (long)wt * a.w_st;
    }
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int i = 0; i < 4; ++i)
        bv[nt][i] = (wrow[nt] >= 0 && koff[i] >= 0) ? W[wrow[nt] + koff[i]] : 0.f;
    if (++ld_ks == ksteps) { ld_ks = 0; ++ld_tap; }
  };

  float a_cur[4], b_cur[NT][4], a_nxt[4], b_nxt[NT][4];
  load_step(a_cur, b_cur);
  for (int s = 0; s < nsteps; ++s) {
    const bool more = (s + 1 < nsteps);
    if (more) load_step(a_nxt, b_nxt);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) acc[nt] = uir_mfma(a_cur[i], b_cur[nt][i], acc[nt]);
    if (more) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        a_cur[i] = a_nxt[i];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) b_cur[nt][i] = b_nxt[nt][i];
      }
    }
  }

  // ---- epilogue: lane = output channel, 16 accumulator registers = 16 pixels ----
  float* __restrict__ Y = a.Y + (long)z * a.y_bs;
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int n = n0 + 32 * nt + r;
    if (n >= a.N) continue;
    const int nr = uir_colmap(n, a.n_grp_real, a.n_grp_pad);
    const float bias = (a.bias && nr >= 0) ? a.bias[nr] : 0.f;
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
      const int mm = m0 + uir_mfma_row(reg, h2);
      if (mm >= a.M) continue;
      float v = acc[nt][reg] + bias;
      if (a.act == UIR_ACT_LEAKY) v = v > 0.f ? v : v * a.slope;
      if (a.epi == UIR_EPI_PLAIN) {
        if (a.R) v += a.R[(long)z * a.r_bs + (long)mm * a.ldr + n];
        Y[(long)mm * a.ldy + n] = v;
      } else if (a.epi == UIR_EPI_GATE_BWD) {
        const float* gp = a.G + (long)z * a.g_bs + (long)mm * a.ldg + n;
        const float g1 = gp[0], g2 = gp[a.g_off];
        Y[(long)mm * a.ldy + n] = v * g2 * uir_gelu_grad(g1);
        Y[(long)mm * a.ldy + a.g_off + n] = v * uir_gelu(g1);
      } else {  // UIR_EPI_LEAKY_BWD: G holds the forward LeakyReLU output
        const float o = a.G[(long)z * a.g_bs + (long)mm * a.ldg + n];
        Y[(long)mm * a.ldy + n] = o > 0.f ? v : v * a.slope;
      }
    }
  }
}

}  // namespace

extern "C" int uir_gemm(const UirGemmArgs* p, void* stream) {
  const UirGemmArgs& a = *p;
  UIR_REQUIRE(a.X && a.W && a.Y);
  UIR_REQUIRE(a.M > 0 && a.N > 0 && a.K > 0 && a.batches > 0);
  UIR_REQUIRE(a.taps == 1 || a.taps == 9);
  UIR_REQUIRE(a.taps == 1 || (a.H > 0 && a.Wd > 0 && a.M % (a.H * a.Wd) == 0));
  UIR_REQUIRE((a.ldx & 3) == 0 && uir_aligned16(a.X) && (a.x_bs & 3) == 0);
  UIR_REQUIRE(a.ldx >= ((a.K + 3) & ~3));  // A rows are read in float4 units up to roundup4(K)
  UIR_REQUIRE(a.ldy >= a.N);
  if (a.pro == UIR_PRO_LN) UIR_REQUIRE((a.K & 3) == 0 && a.ln_g && a.ln_b && a.taps == 1 && uir_aligned16(a.ln_g) && uir_aligned16(a.ln_b));
  if (a.pro == UIR_PRO_GATE) UIR_REQUIRE((a.gate_off & 3) == 0 && a.ldx >= a.gate_off + ((a.K + 3) & ~3));
  if (a.epi == UIR_EPI_GATE_BWD) UIR_REQUIRE(a.G && a.ldg >= a.g_off + a.N && a.ldy >= a.g_off + a.N && !a.R);
  if (a.epi == UIR_EPI_LEAKY_BWD) UIR_REQUIRE(a.G && a.ldg >= a.N && !a.R);
  if (a.R) UIR_REQUIRE(a.ldr >= a.N);
  const int tiles = (a.N + 31) / 32;
  const int groups = (tiles + 3) / 4;
  const int nt = (tiles + groups - 1) / groups;
  dim3 grid((a.M + 127) / 128, groups, a.batches), block(256);
  hipStream_t st = static_cast<hipStream_t>(stream);
  switch (nt) {
    case 1: hipLaunchKernelGGL(uir_gemm_kernel<1>, grid, block, 0, st, a); break;
    case 2: hipLaunchKernelGGL(uir_gemm_kernel<2>, grid, block, 0, st, a); break;
    case 3: hipLaunchKernelGGL(uir_gemm_kernel<3>, grid, block, 0, st, a); break;
    default: hipLaunchKernelGGL(uir_gemm_kernel<4>, grid, block, 0, st, a); break;
  }
  return uir_launch_status();
}
