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
#include <cstdlib>

int uir_conv3x3_mfma(const UirGemmArgs& a, void* stream);      // csrc/conv3m.hip; -1 = not its shape
int uir_conv3x3_small(const UirGemmArgs& a, void* stream);     // csrc/conv3s.hip; -1 = not its shape
int uir_ts_gemm(const UirGemmArgs& a, void* stream);           // csrc/tsgemm.hip; -1 = not its shape

namespace {

// UirGemmArgs.il16: column of hidden channel n of the GDFN tensors in the chunk-interleaved layout (x2 partner 16 columns on)
__device__ __forceinline__ int uir_il16(int n, int il16) { return il16 ? (((n >> 4) << 5) | (n & 15)) : n; }

template <int NT>
__device__ __forceinline__ void uir_gemm_epilogue(const UirGemmArgs& a, const f32x16 (&acc)[NT], int z, int m0, int n0, int r, int h2) {
  // lane = output channel, 16 accumulator registers = 16 pixels
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
      if (a.act == UIR_ACT_GELU) {
        if (a.pre_out) a.pre_out[(long)mm * a.ldpre + n] = v;
        v = uir_gelu(v);
      }
      if (a.epi == UIR_EPI_PLAIN) {
        if (a.out_scale) v *= a.out_scale[mm / a.scale_rows];
        if (a.R) v += a.R[(long)z * a.r_bs + (long)mm * a.ldr + n];
        Y[(long)mm * a.ldy + n] = v;
      } else if (a.epi == UIR_EPI_GATE_BWD) {
        const int nc = uir_il16(n, a.il16);
        const float* gp = a.G + (long)z * a.g_bs + (long)mm * a.ldg + nc;
        const float g1 = gp[0], g2 = gp[a.g_off];
        float ge, dge;
        uir_gelu_pair(g1, ge, dge);
        Y[(long)mm * a.ldy + nc] = v * g2 * dge;
        Y[(long)mm * a.ldy + a.g_off + nc] = v * ge;
      } else {  // UIR_EPI_LEAKY_BWD: G holds the forward LeakyReLU output
        const float o = a.G[(long)z * a.g_bs + (long)mm * a.ldg + n];
        Y[(long)mm * a.ldy + n] = o > 0.f ? v : v * a.slope;
      }
    }
  }
}

template <int NT>
__global__ __launch_bounds__(256) void uir_gemm_kernel(const UirGemmArgs a, const int vecw) {
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
    if (vecw) {  // k-contiguous weight rows (w_sk == 1, K % 4 == 0): one 16-B load per column tile
      const bool kok = kc < a.K;
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        float4 w = make_float4(0.f, 0.f, 0.f, 0.f);
        if (kok && wrow[nt] >= 0) w = *reinterpret_cast<const float4*>(W + wrow[nt] + kc);
        bv[nt][0] = w.x; bv[nt][1] = w.y; bv[nt][2] = w.z; bv[nt][3] = w.w;
      }
      if (++ld_ks == ksteps) { ld_ks = 0; ++ld_tap; }
      return;
    }
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

  uir_gemm_epilogue<NT>(a, acc, z, m0, n0, r, h2);
}

// ---------------------------------------------------------------------------------------
// uir_gemm1x1_kernel: the hot 1x1 path (persistent workgroups).
//  * the (32*NT) x K weight panel (+ LayerNorm gamma/beta) is staged ONCE per workgroup into LDS
//    as [col][Kp] with Kp/4 odd: the per-step ds_read_b128 of a column's 4 consecutive k is
//    bank-conflict free, and LDS reads retire on lgkmcnt, not vmcnt;
//  * A is read from HBM one 32-wide K chunk at a time (4 float4 per lane) and the NEXT unit
//    (next chunk, or chunk 0 of the wave's next row tile) is issued BEFORE this unit's MFMAs
//    and epilogue stores: vmcnt retires in issue order on CDNA4, so the wait for the prefetched
//    loads never waits for the younger stores (measured: 64 % of wave cycles were s_waitcnt);
//  * the epilogue is specialised at compile time (no per-element scalar branches).
// residual rows of one tile, loaded BEFORE the MFMA block so their latency is covered (clamped rows /
// columns: values outside the tile are never used)
template <int NT>
__device__ __forceinline__ void uir_gemm1x1_load_res(const UirGemmArgs& a, float (&rv)[NT][16], int z, int m0, int n0, int r, int h2) {
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int n = min(n0 + 32 * nt + r, a.N - 1);
    const float* rp = a.R + (long)z * a.r_bs + n;
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
      const int mm = min(m0 + 4 * h2 + (reg & 3) + 8 * (reg >> 2), a.M - 1);
      rv[nt][reg] = rp[(long)mm * a.ldr];
    }
  }
}

template <int NT, int EPI, bool HAS_R>
__device__ __forceinline__ void uir_gemm1x1_store(const UirGemmArgs& a, const f32x16 (&acc)[NT], const float (&biasv)[NT],
                                                  const float (&rv)[HAS_R ? NT : 1][16], int z, int m0, int n0, int r, int h2) {
  float* __restrict__ Y = a.Y + (long)z * a.y_bs;
  const float slope = (a.act == UIR_ACT_LEAKY) ? a.slope : 1.0f;
  const bool gelu = (a.act == UIR_ACT_GELU);
  const bool full = (m0 + 32 <= a.M);
  // per-sample output factor: a 32-row tile touches at most two samples (scale_rows >= 32) - one division per tile, not per element
  // (only with a residual - its one use - so that the variants without one, the hot ones of the SpectralTransformer step, carry nothing)
  float sc0 = 1.f, sc1 = 1.f;
  int sc_bnd = 0;
  if constexpr (HAS_R) {
    if (a.out_scale) {
      const int s0 = m0 / a.scale_rows;
      sc_bnd = (s0 + 1) * a.scale_rows;
      sc0 = a.out_scale[s0];
      sc1 = a.out_scale[sc_bnd < a.M ? s0 + 1 : s0];
    }
  }
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int n = n0 + 32 * nt + r;
    if (n >= a.N) continue;
    const float bias = biasv[nt];
    const long row0 = (long)m0 + 4 * h2;
    const int nc = (EPI == UIR_EPI_GATE_BWD) ? uir_il16(n, a.il16) : n;      // column of hidden channel n in the GDFN tensors
    float* yp = Y + row0 * a.ldy + nc;
    const float* gp = (EPI != UIR_EPI_PLAIN) ? a.G + (long)z * a.g_bs + row0 * a.ldg + nc : nullptr;
    float* pp = (EPI == UIR_EPI_PLAIN && gelu && a.pre_out) ? a.pre_out + row0 * a.ldpre + n : nullptr;
    auto put = [&](int reg) {
      const int dr = (reg & 3) + 8 * (reg >> 2);   // row offset inside the tile (compile-time)
      float v = acc[nt][reg] + bias;
      v = v > 0.f ? v : v * slope;
      if constexpr (EPI == UIR_EPI_PLAIN) {
        if (gelu) {                                // uniform over the launch
          if (pp) pp[(long)dr * a.ldpre] = v;
          v = uir_gelu(v);
        }
        if constexpr (HAS_R) {
          if (a.out_scale) v *= ((int)row0 + dr < sc_bnd) ? sc0 : sc1;                        // uniform over the launch
          v += rv[nt][reg];
        }
        yp[(long)dr * a.ldy] = v;
      } else if constexpr (EPI == UIR_EPI_GATE_BWD) {
        const float g1 = gp[(long)dr * a.ldg], g2 = gp[(long)dr * a.ldg + a.g_off];
        float ge, dge;
        uir_gelu_pair(g1, ge, dge);
        yp[(long)dr * a.ldy] = v * g2 * dge;
        yp[(long)dr * a.ldy + a.g_off] = v * ge;
      } else {
        const float o = gp[(long)dr * a.ldg];
        yp[(long)dr * a.ldy] = o > 0.f ? v : v * a.slope;
      }
    };
    if (full) {
      if constexpr (EPI == UIR_EPI_PLAIN) {
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) put(reg);
      } else {
        // All operand loads of the tile first, then compute + store.  Written load/compute/store per register the
        // compiler must keep every load behind the previous stores (G may alias Y for all it knows): 16 serialized
        // HBM round trips per 32x32 tile (ISA: [global_load x2, s_waitcnt vmcnt(0), global_store x2] x16).
        float g1[16], g2[16];
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
          const long o = (long)((reg & 3) + 8 * (reg >> 2)) * a.ldg;
          g1[reg] = gp[o];
          if constexpr (EPI == UIR_EPI_GATE_BWD) g2[reg] = gp[o + a.g_off];
        }
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
          const long o = (long)((reg & 3) + 8 * (reg >> 2)) * a.ldy;
          float v = acc[nt][reg] + bias;
          v = v > 0.f ? v : v * slope;
          if constexpr (EPI == UIR_EPI_GATE_BWD) {
            float ge, dge;
            uir_gelu_pair(g1[reg], ge, dge);
            yp[o] = v * g2[reg] * dge;
            yp[o + a.g_off] = v * ge;
          } else {
            yp[o] = g1[reg] > 0.f ? v : v * a.slope;
          }
        }
      }
    } else {
#pragma unroll
      for (int reg = 0; reg < 16; ++reg)
        if (m0 + 4 * h2 + (reg & 3) + 8 * (reg >> 2) < a.M) put(reg);
    }
  }
}

// GELU-backward epilogue with the bias gradient (EPI_GELU_BWD; LeFF's Linear2 input gradient, src/model/block.py:240-246):
//   out = acc * gelu'(G + bias),  db[n] += out  - the gradient of the depthwise output u and of the depthwise bias, given
//   G = u and the gradient acc of gelu(u + bias).  One GELU evaluation per element, where the accumulator already sits.
template <int NT>
__device__ __forceinline__ void uir_gemm1x1_store_gelu_bwd(const UirGemmArgs& a, const f32x16 (&acc)[NT], const float (&biasv)[NT],
                                                            float (&dbacc)[NT], int z, int m0, int n0, int r, int h2) {
  const bool full = (m0 + 32 <= a.M);
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int n = n0 + 32 * nt + r;
    if (n >= a.N) continue;
    const long row0 = (long)m0 + 4 * h2;
    float* yp = a.Y + (long)z * a.y_bs + row0 * a.ldy + n;
    const float* gp = a.G + (long)z * a.g_bs + n;
    float g1[16];
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {                       // all operand loads first (see uir_gemm1x1_store)
      const long row = row0 + (reg & 3) + 8 * (reg >> 2);
      g1[reg] = gp[(full || row < a.M ? row : (long)a.M - 1) * a.ldg];
    }
    float s = 0.f;
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
      const int dr = (reg & 3) + 8 * (reg >> 2);
      const float o = acc[nt][reg] * uir_gelu_grad(g1[reg] + biasv[nt]);
      if (full || row0 + dr < a.M) {
        yp[(long)dr * a.ldy] = o;
        s += o;
      }
    }
    dbacc[nt] += s;
  }
}

// Gate-backward epilogue that ALSO accumulates the project_out weight gradient (EPI_GATE_BWD_DW, K <= 32, full tiles):
//   dW_out[c][j] = sum_px dY[px][c] * g[px][j],   g = gelu(t2a) * t2b  (computed here anyway).
// The accumulator layout of the main MFMA (lane = hidden column j, register = pixel) IS the B-operand layout of a
// contraction over pixels, so g feeds the second MFMA straight from registers; the A operand dY^T (lane = channel)
// is 16 coalesced dword loads of rows this wave has just read.  gacc lives across the workgroup's row tiles.
template <int NT>
__device__ __forceinline__ void uir_gemm1x1_store_gate_dw(const UirGemmArgs& a, const f32x16 (&acc)[NT], f32x16 (&gacc)[NT],
                                                           int m0, int n0, int r, int h2) {
  float ap[16];
  {
    const int kc = r < a.K ? r : 0;
    const float* xp = a.X + ((long)m0 + 4 * h2) * a.ldx + kc;
#pragma unroll
    for (int s = 0; s < 16; ++s) ap[s] = xp[(long)((s & 3) + 8 * (s >> 2)) * a.ldx];
    if (r >= a.K) {
#pragma unroll
      for (int s = 0; s < 16; ++s) ap[s] = 0.f;
    }
  }
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int n = n0 + 32 * nt + r;
    const bool nok = n < a.N;
    const int nn = uir_il16(nok ? n : a.N - 1, a.il16);  // clamped: no lane-divergent control flow around the MFMAs
    const long row0 = (long)m0 + 4 * h2;
    float* yp = a.Y + row0 * a.ldy + nn;
    const float* gp = a.G + row0 * a.ldg + nn;
    float g1[16], g2[16];
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
      const long o = (long)((reg & 3) + 8 * (reg >> 2)) * a.ldg;
      g1[reg] = gp[o];
      g2[reg] = gp[o + a.g_off];
    }
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
      const long o = (long)((reg & 3) + 8 * (reg >> 2)) * a.ldy;
      const float v = acc[nt][reg];
      float ge, dge;
      uir_gelu_pair(g1[reg], ge, dge);
      if (nok) {
        yp[o] = v * g2[reg] * dge;
        yp[o + a.g_off] = v * ge;
      }
      g1[reg] = nok ? ge * g2[reg] : 0.f;                // g, in B-operand position
    }
#pragma unroll
    for (int s = 0; s < 16; ++s) gacc[nt] = uir_mfma(ap[s], g1[s], gacc[nt]);
  }
}

// Plain store + fused per-batch Gram (EPI_PLAIN_DW, K <= 32, one column group, full tiles): dW[k][j] += sum_px X[px][k] * Z[px][j] over the
// tile's pixels.  A operand X^T (lane = channel k) as in uir_gemm1x1_store_gate_dw; B operand Z = a.G (lane = column j, register = pixel):
// 16 dword loads of whole rows.  One 32 x 32 dW tile (Z has <= 32 columns), accumulated across the workgroup's row tiles.
template <int NT>
__device__ __forceinline__ void uir_gemm1x1_store_plain_dw(const UirGemmArgs& a, const f32x16 (&acc)[NT], f32x16& gacc, int z, int m0, int n0,
                                                            int r, int h2) {
  const long row0 = (long)m0 + 4 * h2;
  const bool kok = r < a.K, jok = r < a.dw_cols;
  float ap[16], zv[16];
  {
    const float* xp = a.X + (long)z * a.x_bs + row0 * a.ldx + (kok ? r : 0);
    const float* zp = a.G + (long)z * a.g_bs + row0 * a.ldg + (jok ? r : 0);
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      const long o = (long)((s & 3) + 8 * (s >> 2));
      ap[s] = xp[o * a.ldx];
      zv[s] = zp[o * a.ldg];
    }
  }
  float* Y = a.Y + (long)z * a.y_bs;
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int n = n0 + 32 * nt + r;
    if (n < a.N) {
      float* yp = Y + row0 * a.ldy + n;
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) yp[(long)((reg & 3) + 8 * (reg >> 2)) * a.ldy] = acc[nt][reg];
    }
  }
#pragma unroll
  for (int s = 0; s < 16; ++s) gacc = uir_mfma(kok ? ap[s] : 0.f, jok ? zv[s] : 0.f, gacc);
}

template <int NT, int PRO, int EPI, bool HAS_R>
__global__ __launch_bounds__(256, EPI == UIR_EPI_GATE_BWD_DW ? 2 : 3) void uir_gemm1x1_kernel(const UirGemmArgs a, const int Kp, const int tiles) {
  extern __shared__ __attribute__((aligned(16))) float wl[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 31, h2 = lane >> 5;
  const int z = blockIdx.z;
  const int n0 = blockIdx.y * (32 * NT);
  const float* __restrict__ X = a.X + (long)z * a.x_bs;
  const float* __restrict__ W = a.W + (long)z * a.w_bs;
  const int K8 = (a.K + 7) & ~7;
  constexpr int NC = 32 * NT;
  const int cpt = (K8 + 31) >> 5;          // 32-wide chunks per tap
  const int ntap = a.ntap > 0 ? a.ntap : a.taps;   // taps [tap0, tap0 + ntap) are contracted (panel slots 0..ntap-1)
  const int nchunks = cpt * ntap;          // units per row tile
  const int HW = a.H * a.Wd;
  // Branch-free loads (a predicated load costs a branch + a full vmcnt(0) drain in hipcc's
  // codegen): rows past M are clamped to the last row (their results are never stored) and k
  // past K to column 0 (they meet exact-zero weights staged in LDS), so every address is valid.
  // Two statically named register panels (P0/P1) alternate between "being consumed" and
  // "being filled": the loop is unrolled by two so no register copies (and no vmcnt(0) at the
  // back-edge, behind the epilogue stores) are ever needed.
  constexpr int NU = (PRO == UIR_PRO_GATE) ? 4 : 1;
  float4 P0[4], P1[4], U0[NU], U1[NU];
  const int kmax = (a.K + 3) & ~3;
  // returns whether this lane's (row, tap) neighbour exists (else the loaded values must be zeroed)
  int ct_tile = -1, ct_ph = 0, ct_pw = 0;   // 3x3: this lane's pixel coordinates, divided out once per row tile (not per unit)
  auto load_unit = [&](float4 (&an)[4], float4 (&un)[NU], int tile_, int c_) -> bool {
    const int mm = min(tile_ * 128 + wave * 32 + r, a.M - 1);
    long row = mm;
    bool ok = true;
    int cc = c_;
    if (a.taps > 1) {
      const int slot = c_ / cpt;
      cc = c_ - slot * cpt;
      const int tap = a.tap0 + slot;
      if (tile_ != ct_tile) {
        const int hw = mm % HW;
        ct_ph = hw / a.Wd;
        ct_pw = hw - ct_ph * a.Wd;
        ct_tile = tile_;
      }
      const int ph = ct_ph, pw = ct_pw;
      const int dy = tap / 3 - 1, dx = tap - (tap / 3) * 3 - 1;
      const int hh = ph + dy, ww = pw + dx;
      ok = hh >= 0 && hh < a.H && ww >= 0 && ww < a.Wd;
      row = ok ? (long)mm + dy * a.Wd + dx : (long)mm;
    }
    const float* xr = X + row * a.ldx;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      int kc = 32 * cc + 8 * s + 4 * h2;
      kc = (kc < kmax) ? kc : 0;
      an[s] = *reinterpret_cast<const float4*>(xr + kc);
      if constexpr (PRO == UIR_PRO_GATE) un[s] = *reinterpret_cast<const float4*>(xr + a.gate_off + kc);
    }
    return ok;
  };

  // first A panel in flight before (and during) the weight staging below
  bool ok0 = true, ok1 = true;
  if ((int)blockIdx.x < tiles) ok0 = load_unit(P0, U0, blockIdx.x, 0);

  // panel [col][tap][K8] (+pad): element (n, k, tap) -> wl[nl*Kp + tap*K8 + k]; tap-flipped for
  // input-gradients.  Row/column weight offsets are tabulated once per workgroup (the colmap
  // divisions cost ~30 VALU ops each: done NC + K8 times, not NC*K8 times) and the panel is walked
  // with 2-D loops whose fast index follows the contiguous weight dimension.
  float* lng = wl + NC * Kp;                              // gamma[K8], beta[K8] (LN prologue only)
  int* nmap = reinterpret_cast<int*>(lng + 2 * K8);       // [NC]  weight offset of column nl, or -1
  int* kmap = nmap + NC;                                  // [K8]  weight offset of k, or -1
  // per-wave transpose buffer of the float4 epilogue (16-byte aligned: everything before it is a multiple of 4 floats)
  for (int i = threadIdx.x; i < NC + K8; i += 256) {
    if (i < NC) {
      const int n = n0 + i;
      const int nr = (n < a.N) ? uir_colmap(n, a.n_grp_real, a.n_grp_pad) : -1;
      nmap[i] = nr < 0 ? -1 : nr * (int)a.w_sn;
    } else {
      const int k = i - NC;
      const int kr = (k < a.K) ? uir_colmap(k, a.k_grp_real, a.k_grp_pad) : -1;
      kmap[k] = kr < 0 ? -1 : kr * (int)a.w_sk;
    }
  }
  if constexpr (PRO == UIR_PRO_LN) {
    for (int k = threadIdx.x; k < K8; k += 256) {
      lng[k] = k < a.K ? a.ln_g[k] : 0.f;
      lng[K8 + k] = k < a.K ? a.ln_b[k] : 0.f;
    }
  }
  __syncthreads();
  {
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
    for (int tap = 0; tap < ntap; ++tap) {
      const int wt = (a.flip ? (a.taps - 1 - (a.tap0 + tap)) : (a.tap0 + tap)) * (int)a.w_st;
      float* wp = wl + tap * K8;
      // Loads are issued in batches of up to 16 per thread before any of them is stored: one load per loop trip (the first
      // version) serialised ~90 L2 round trips per workgroup - 25 us of the 65 us of a K = 704 launch at the deep levels, where
      // a workgroup stages its panel for ONE row tile.
      if (a.w_sk == 1) {          // k contiguous in memory: lanes along k; 4 rows x 4 k-groups per trip
        for (int nl0 = ty; nl0 < NC; nl0 += 32) {
          for (int k0 = tx; k0 < K8; k0 += 128) {
            float v[4][4];
            bool ok[4][4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
              const int nl = nl0 + 8 * u;
              const int no = nl < NC ? nmap[nl] : -1;
#pragma unroll
              for (int w = 0; w < 4; ++w) {
                const int k = k0 + 32 * w;
                const int ko = k < K8 ? kmap[k] : -1;
                ok[u][w] = no >= 0 && ko >= 0;
                v[u][w] = W[ok[u][w] ? no + ko + wt : 0];
              }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
              for (int w = 0; w < 4; ++w)
                if (nl0 + 8 * u < NC && k0 + 32 * w < K8) wp[(nl0 + 8 * u) * Kp + k0 + 32 * w] = ok[u][w] ? v[u][w] : 0.f;
          }
        }
      } else {                    // n (or neither) contiguous: lanes along n; 8 k x NT columns per trip
        for (int k0 = ty; k0 < K8; k0 += 64) {
          float v[8][NT];
          bool ok[8][NT];
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            const int k = k0 + 8 * u;
            const int ko = k < K8 ? kmap[k] : -1;
#pragma unroll
            for (int m = 0; m < NT; ++m) {
              const int no = nmap[tx + 32 * m];
              ok[u][m] = no >= 0 && ko >= 0;
              v[u][m] = W[ok[u][m] ? no + ko + wt : 0];
            }
          }
#pragma unroll
          for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int m = 0; m < NT; ++m)
              if (k0 + 8 * u < K8) wp[(tx + 32 * m) * Kp + k0 + 8 * u] = ok[u][m] ? v[u][m] : 0.f;
        }
      }
    }
  }
  __syncthreads();
  const float* wbase = wl + r * Kp + 4 * h2;
  float bias[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int n = n0 + 32 * nt + r;
    const int nr = (n < a.N) ? uir_colmap(n, a.n_grp_real, a.n_grp_pad) : -1;
    bias[nt] = (a.bias && nr >= 0) ? a.bias[nr] : 0.f;
  }

  int tile = blockIdx.x, c = 0;
  float mean = 0.f, rstd = 0.f;
  f32x16 acc[NT];
  f32x16 gacc[EPI == UIR_EPI_GATE_BWD_DW ? NT : 1];   // fused weight-gradient accumulators (live across row tiles)
  float dbacc[EPI == UIR_EPI_GELU_BWD ? NT : 1];       // fused bias-gradient sums (column = lane)
#pragma unroll
  for (int nt = 0; nt < (EPI == UIR_EPI_GELU_BWD ? NT : 1); ++nt) dbacc[nt] = 0.f;
  if constexpr (EPI == UIR_EPI_GATE_BWD_DW) {
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int q = 0; q < 16; ++q) gacc[nt][q] = 0.f;
  }
  if constexpr (EPI == UIR_EPI_PLAIN_DW) {
#pragma unroll
    for (int q = 0; q < 16; ++q) gacc[0][q] = 0.f;
  }

  auto step = [&](float4 (&an)[4], float4 (&un)[NU], const bool anok, float4 (&nx)[4], float4 (&nu)[NU], bool& nxok) -> bool {
    const int m0 = tile * 128 + wave * 32;
    const int m = m0 + r;
    if (c == 0) {
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[nt][i] = 0.f;
      if constexpr (PRO == UIR_PRO_LN) {
        float s = 0.f, q = 0.f;
        if (nchunks == 1) {  // the whole row is already in registers
#pragma unroll
          for (int t = 0; t < 4; ++t) s += (8 * t + 4 * h2 < a.K) ? (an[t].x + an[t].y) + (an[t].z + an[t].w) : 0.f;
          s += __shfl_xor(s, 32);
          mean = s / (float)a.K;
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            const float d0 = an[t].x - mean, d1 = an[t].y - mean, d2 = an[t].z - mean, d3 = an[t].w - mean;
            q += (8 * t + 4 * h2 < a.K) ? (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3) : 0.f;
          }
        } else {
          const float* xr = X + (long)min(m, a.M - 1) * a.ldx;
          for (int kc = 4 * h2; kc < a.K; kc += 8) {
            const float4 v = *reinterpret_cast<const float4*>(xr + kc);
            s += (v.x + v.y) + (v.z + v.w);
          }
          s += __shfl_xor(s, 32);
          mean = s / (float)a.K;
          for (int kc = 4 * h2; kc < a.K; kc += 8) {
            const float4 v = *reinterpret_cast<const float4*>(xr + kc);
            const float d0 = v.x - mean, d1 = v.y - mean, d2 = v.z - mean, d3 = v.w - mean;
            q += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
          }
        }
        q += __shfl_xor(q, 32);
        rstd = rsqrtf(q / (float)a.K + a.ln_eps);
        if (a.stats && blockIdx.y == 0 && h2 == 0 && m < a.M)
          *reinterpret_cast<float2*>(a.stats + ((long)z * a.M + m) * 2) = make_float2(mean, rstd);
      }
    }
    const int tapc = (a.taps > 1) ? c / cpt : 0;
    const int cin = c - tapc * cpt;           // chunk inside the tap
    float4 ac[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      float4 v = anok ? an[s] : make_float4(0.f, 0.f, 0.f, 0.f);
      const int kc = 32 * cin + 8 * s + 4 * h2;
      if constexpr (PRO == UIR_PRO_LN) {
        const int kq = (kc < K8) ? kc : 0;
        const float4 g = *reinterpret_cast<const float4*>(lng + kq);
        const float4 b = *reinterpret_cast<const float4*>(lng + K8 + kq);
        v.x = (v.x - mean) * rstd * g.x + b.x;
        v.y = (v.y - mean) * rstd * g.y + b.y;
        v.z = (v.z - mean) * rstd * g.z + b.z;
        v.w = (v.w - mean) * rstd * g.w + b.w;
      } else if constexpr (PRO == UIR_PRO_GATE) {
        const float4 u = un[s];
        v.x = uir_gelu(v.x) * u.x;
        v.y = uir_gelu(v.y) * u.y;
        v.z = uir_gelu(v.z) * u.z;
        v.w = uir_gelu(v.w) * u.w;
      }
      ac[s] = v;
    }
    // next unit in flight before this unit's MFMAs (and, for the last chunk, before the stores)
    const bool last = (c + 1 == nchunks);
    const int tn = last ? tile + (int)gridDim.x : tile;
    const int cn = last ? 0 : c + 1;
    nxok = load_unit(nx, nu, tn, cn);
    float rv[HAS_R ? NT : 1][16];
    if constexpr (HAS_R) {
      if (last) uir_gemm1x1_load_res<NT>(a, rv, z, m0, n0, r, h2);
    }
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int k0 = 32 * cin + 8 * s;
      if (k0 < K8) {
        float4 b[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) b[nt] = *reinterpret_cast<const float4*>(wbase + nt * 32 * Kp + tapc * K8 + k0);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          acc[nt] = uir_mfma(ac[s].x, b[nt].x, acc[nt]);
          acc[nt] = uir_mfma(ac[s].y, b[nt].y, acc[nt]);
          acc[nt] = uir_mfma(ac[s].z, b[nt].z, acc[nt]);
          acc[nt] = uir_mfma(ac[s].w, b[nt].w, acc[nt]);
        }
      }
    }
    if (last && m0 < a.M) {
      if constexpr (EPI == UIR_EPI_GATE_BWD_DW) uir_gemm1x1_store_gate_dw<NT>(a, acc, gacc, m0, n0, r, h2);
      else if constexpr (EPI == UIR_EPI_PLAIN_DW) uir_gemm1x1_store_plain_dw<NT>(a, acc, gacc[0], z, m0, n0, r, h2);
      else if constexpr (EPI == UIR_EPI_GELU_BWD) uir_gemm1x1_store_gelu_bwd<NT>(a, acc, bias, dbacc, z, m0, n0, r, h2);
      else uir_gemm1x1_store<NT, EPI, HAS_R>(a, acc, bias, rv, z, m0, n0, r, h2);
    }
    tile = tn;
    c = cn;
    return tile < tiles;
  };

  if (tile < tiles) {
    while (true) {
      if (!step(P0, U0, ok0, P1, U1, ok1)) break;
      if (!step(P1, U1, ok1, P0, U0, ok0)) break;
    }
  }
  if constexpr (EPI == UIR_EPI_GELU_BWD) {
    // bias-gradient partials of this workgroup's columns: lane halves, then the 4 waves in a fixed order, one slab row
    __syncthreads();
    float* red = wl;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const float g = dbacc[nt] + __shfl_xor(dbacc[nt], 32);
      if (h2 == 0) red[wave * (NT * 32) + nt * 32 + r] = g;
    }
    __syncthreads();
    float* slab = a.dw_slab + (long)blockIdx.x * a.dw_stride;
    for (int e = threadIdx.x; e < NT * 32; e += 256) {
      const int n = n0 + e;
      if (n < a.N) slab[n] = (red[e] + red[NT * 32 + e]) + (red[2 * NT * 32 + e] + red[3 * NT * 32 + e]);
    }
  }
  if constexpr (EPI == UIR_EPI_PLAIN_DW) {
    // the 4 waves' partial dW tiles meet in LDS in a fixed order (the weight panel is dead); one slab row per workgroup, batch z's columns
    __syncthreads();
    float* red = wl;
    for (int w = 0; w < 4; ++w) {
      if (wave == w) {
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          float* p = &red[q * 64 + lane];
          *p = (w == 0) ? gacc[0][q] : (*p + gacc[0][q]);
        }
      }
      __syncthreads();
    }
    float* slab = a.dw_slab + (long)blockIdx.x * a.dw_stride + (long)z * a.K * a.dw_cols;
    for (int e = threadIdx.x; e < 1024; e += 256) {
      const int ln = e & 63, q = e >> 6;
      const int cch = uir_mfma_row(q, ln >> 5), j = ln & 31;
      if (cch < a.K && j < a.dw_cols) slab[(long)cch * a.dw_cols + j] = red[e];
    }
  }
  if constexpr (EPI == UIR_EPI_GATE_BWD_DW) {
    // sum the 4 waves' partial dW tiles through LDS in a fixed order (the weight panel is dead now) and write this
    // workgroup's slab row; uir_reduce_slab sums the rows: deterministic, no atomics
    __syncthreads();
    float* red = wl;
    for (int w = 0; w < 4; ++w) {
      if (wave == w) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
          for (int q = 0; q < 16; ++q) {
            float* p = &red[(nt * 16 + q) * 64 + lane];
            *p = (w == 0) ? gacc[nt][q] : (*p + gacc[nt][q]);
          }
      }
      __syncthreads();
    }
    float* slab = a.dw_slab + (long)blockIdx.x * a.dw_stride;
    const int hreal = a.n_grp_pad > 0 ? a.n_grp_real : a.N;   // real hidden width (columns n_grp_real.. are pads)
    for (int e = threadIdx.x; e < NT * 1024; e += 256) {
      const int ln = e & 63, q = (e >> 6) & 15, nt = e >> 10;
      const int cch = uir_mfma_row(q, ln >> 5);               // dY channel = row of the dW tile
      const int j = n0 + 32 * nt + (ln & 31);                 // hidden column
      if (cch < a.K && j < hreal) slab[(long)cch * hreal + j] = red[e];
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
  UIR_REQUIRE(a.il16 == 0 || (a.il16 == 1 && (a.epi == UIR_EPI_GATE_BWD || a.epi == UIR_EPI_GATE_BWD_DW) && a.g_off == 16 &&
                              (a.N & 15) == 0 && a.ldg >= 2 * a.N && a.ldy >= 2 * a.N));
  if (a.epi == UIR_EPI_GATE_BWD) UIR_REQUIRE(a.G && (a.il16 || (a.ldg >= a.g_off + a.N && a.ldy >= a.g_off + a.N)) && !a.R);
  if (a.epi == UIR_EPI_LEAKY_BWD) UIR_REQUIRE(a.G && a.ldg >= a.N && !a.R);
  UIR_REQUIRE(a.epi >= UIR_EPI_PLAIN && a.epi <= UIR_EPI_LN_BWD);
  if (a.epi == UIR_EPI_PLAIN_DW) {
    // plain store + fused per-batch Gram against G: K <= 32, one column group, whole 128-row tiles, no bias / activation / residual
    UIR_REQUIRE(a.G && a.dw_slab && a.dw_cols > 0 && a.dw_cols <= 32 && a.ldg >= a.dw_cols && a.K <= 32 && a.N <= 96 && !a.R && !a.bias &&
                a.act == UIR_ACT_NONE && a.taps == 1 && a.pro == UIR_PRO_NONE && (a.M & 127) == 0 && a.n_grp_pad <= 0 && a.k_grp_pad <= 0 &&
                a.dw_stride >= (long)a.batches * a.K * a.dw_cols && a.dw_rows >= 1 && a.dw_rows <= a.M / 128);
  }
  if (a.epi == UIR_EPI_GELU_BWD) {
    // out = acc * gelu'(G + bias), per-workgroup column sums of out to dw_slab rows [gridDim.x][dw_stride]
    UIR_REQUIRE(a.G && a.ldg >= a.N && a.bias && a.dw_slab && a.dw_stride >= a.N && !a.R && a.act == UIR_ACT_NONE);
    UIR_REQUIRE(a.taps == 1 && a.batches == 1 && a.pro == UIR_PRO_NONE && a.n_grp_pad <= 0);
  }
  if (a.epi == UIR_EPI_GATE_BWD_DW) {
    // fused dW_out: one column group (N <= 96), K <= 32 channels, whole 128-row tiles, pads only at the tail (N = hp)
    UIR_REQUIRE(a.G && (a.il16 || (a.ldg >= a.g_off + a.N && a.ldy >= a.g_off + a.N)) && !a.R && !a.bias && a.act == UIR_ACT_NONE);
    UIR_REQUIRE(a.dw_slab && a.dw_stride >= (long)a.K * (a.n_grp_pad > 0 ? a.n_grp_real : a.N) && a.K <= 32 && a.N <= 96 &&
                a.taps == 1 && a.batches == 1);
    UIR_REQUIRE((a.M & 127) == 0 && a.pro == UIR_PRO_NONE && (a.n_grp_pad <= 0 || a.n_grp_pad == a.N));
  }
  if (a.R) UIR_REQUIRE(a.ldr >= a.N);
  if (a.out_scale) UIR_REQUIRE(a.R && a.scale_rows >= 32 && a.epi == UIR_EPI_PLAIN && a.taps == 1 && a.batches == 1 && !a.pre_out);
  if (a.act == UIR_ACT_GELU) UIR_REQUIRE(a.epi == UIR_EPI_PLAIN && (!a.pre_out || (a.batches == 1 && a.ldpre >= a.N)));
  else UIR_REQUIRE(!a.pre_out);
  UIR_REQUIRE(a.ntap <= 0 || (a.taps == 9 && a.tap0 >= 0 && a.tap0 + a.ntap <= 9));
  UIR_REQUIRE(a.ntap > 0 || a.tap0 == 0);
  {
    // 3 x 3 convolutions with <= 32 channels on both sides (outputl / output / down1 / down2 and their input gradients): the direct
    // VALU kernel of csrc/conv3s.hip (an MFMA tile is 3/4 ... 7/8 empty for 8 or 4 output channels)
    // 8 .. 32 channels on both sides: the LDS-staged 16x16x4 MFMA kernel of csrc/conv3m.hip
    const int rm = uir_conv3x3_mfma(a, stream);
    if (rm >= 0) return rm;
    const int rc = uir_conv3x3_small(a, stream);
    if (rc >= 0) return rc;
  }
  {
    // the deep levels' plain contractions (16 k - 64 k rows; per-sample weights, padded K groups and a residual included): split-bf16
    // on 64- / 128-row tiles (csrc/tsgemm.hip)
    const int rc = uir_ts_gemm(a, stream);
    if (rc >= 0) return rc;
    UIR_REQUIRE(a.epi != UIR_EPI_LN_BWD);       // the LayerNorm-backward epilogue exists on that kernel only
  }
  const int tiles = (a.N + 31) / 32;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int ntap = a.ntap > 0 ? a.ntap : a.taps;
  {
    // Wide plain Linear layers (NewBigFRFNModel / AST: K, N up to 2048) on the LDS-tiled kernel of csrc/tiled.hip.  The panel
    // kernel below keeps ALL K of one column group in LDS (fine for the SpectralTransformer's K <= 340), its un-staged fallback
    // ran the K >= 1024 layers at 22-29 TFLOP/s.  UIR_TILED_MINK (default 256): smallest K sent there; 0 disables.
    // UIR_MIXED: the 16-bit kernel has no MFMA time to lose on a short contraction - every Linear with >= 64 input channels goes there
    // (AST's 64 -> 256 / 128 -> 512 LeFF layers on 1 M / 262 k tokens stream at the byte rate instead of 64-81 TFLOP/s of f32 MFMA)
    const int mixed = uir_mixed_mode();
    // f32: K >= 256; K >= 128 where the rows are very many (AST's 128 -> 512 LeFF / projection layers on 262 k tokens: the tiled
    // kernels' split-bf16 form and row-major stores against this file's panel kernel at 81 TFLOP/s)
    const int mink = mixed != 0 ? 64 : (a.M >= 524288 ? 64 : a.M >= 131072 ? 128 : 256);
    const bool w_kc = a.w_sk == 1 && (a.w_sn & 3) == 0, w_nc = a.w_sn == 1 && (a.w_sk & 3) == 0 && (a.N & 3) == 0;
    if (mink > 0 && a.K >= mink && a.N >= (mixed != 0 ? 64 : 128) && a.M >= 256 && a.taps == 1 && a.batches == 1 && a.pro == UIR_PRO_NONE &&
        a.epi == UIR_EPI_PLAIN && (a.act == UIR_ACT_NONE || a.act == UIR_ACT_GELU) && a.n_grp_pad <= 0 && a.k_grp_pad <= 0 &&
        (a.K & 3) == 0 && (w_kc || w_nc) && uir_aligned16(a.W)) {
      UirTiledArgs t{};
      t.A = a.X; t.sa_i = a.ldx; t.sa_k = 1;
      t.B = a.W; t.sb_j = a.w_sn; t.sb_k = a.w_sk;
      t.D = a.Y; t.sd_i = a.ldy; t.sd_j = 1; t.d_split = 0;
      t.MM = a.M; t.NN = a.N; t.KK = a.K; t.k_per_split = a.K;
      t.bias = a.bias; t.R = a.R; t.ldr = a.ldr; t.pre_out = a.pre_out; t.ldpre = a.ldpre; t.act = a.act;
      t.out_scale = a.out_scale; t.scale_rows = a.scale_rows;
      t.allow_split = a.K >= 512 || a.N >= 256;
      return uir_tiled_gemm(t, st);
    }
    // 3 x 3 convolutions with many input channels (sampler convolutions of AST / NewBigFRFNModel, the SpectralTransformer's UpSample1):
    // one launch on the LDS-tiled kernel instead of tap-group launches / the un-staged fallback.  UIR_TILED_CONV_MINK (default 256; 0 off);
    // the SpectralTransformer's own 3 x 3 convolutions (K <= 256 on <= 16 384 pixels, side lane) stay on the tap-group path they were
    // tuned and measured on
    // (round 3: ups_1.Sups itself, K = 128 -> N = 256 forward, 146 us here against 193 us as tap groups; its and ups_2's input
    // gradients - flipped taps, K = 256 / 128 - measured 352 / 240 us here against 226 / 166 us and stay on the tap-group path)
    constexpr int cmink = 256;
    const bool ups1_fwd = a.taps == 9 && !a.flip && a.K >= 128 && a.N >= 256;
    if (cmink > 0 && a.taps == 9 && a.ntap <= 0 && (ups1_fwd || (a.K >= cmink && (a.K >= 512 || a.M > 16384))) && a.N >= 64 && a.batches == 1 && a.pro == UIR_PRO_NONE &&
        a.epi == UIR_EPI_PLAIN && a.act == UIR_ACT_NONE && a.n_grp_pad <= 0 && a.k_grp_pad <= 0 && (a.K & 3) == 0) {
      UirTiledArgs t{};
      t.A = a.X; t.sa_i = a.ldx; t.sa_k = 1;
      t.B = a.W; t.sb_j = a.w_sn; t.sb_k = a.w_sk; t.sb_t = a.w_st;
      t.D = a.Y; t.sd_i = a.ldy; t.sd_j = 1;
      t.MM = a.M; t.NN = a.N; t.KK = a.K; t.k_per_split = a.K;
      t.bias = a.bias; t.R = a.R; t.ldr = a.ldr; t.act = a.act;
      t.H = a.H; t.Wd = a.Wd; t.flip = a.flip;
      return uir_tiled_conv3x3(t, st);
    }
  }
  {
    const int K8 = (a.K + 7) & ~7;
    // A 3x3 convolution whose nine-tap panel of even ONE 32-column tile does not fit the LDS path (K > 80: ups_1.Sups,
    // ups_{1,2}.Sups input gradients, 9.7 GFLOP each) runs as launches over tap groups of <= 50 KB that accumulate into Y;
    // the un-staged fallback below reads weights per MFMA step and reaches 22-29 TFLOP/s there.  (K = 64 keeps its single
    // 74 KB launch: 178 us against 219 us split.)
    if (a.taps == 9 && a.ntap <= 0 && (size_t)32 * (K8 * 9 + 4) * 4 > 96 * 1024 && (size_t)32 * (K8 + 4) * 4 <= 48 * 1024 &&
        a.pro == UIR_PRO_NONE && a.epi == UIR_EPI_PLAIN && a.act == UIR_ACT_NONE) {
      int g = 1;
      while (g < 3 && (size_t)32 * (K8 * (g + 1) + 4) * 4 <= 50 * 1024) ++g;
      if (g == 2) g = 1;                                        // groups of whole kernel rows (3 taps) or single taps
      for (int t0 = 0; t0 < 9; t0 += g) {
        UirGemmArgs b = a;
        b.tap0 = t0;
        b.ntap = g;
        if (t0 > 0) { b.R = a.Y; b.ldr = a.ldy; b.r_bs = a.y_bs; b.bias = nullptr; }
        const int rc = uir_gemm(&b, stream);
        if (rc != UIR_OK) return rc;
      }
      return UIR_OK;
    }
    // LDS-resident weight panel (all contracted taps): pick the widest column group (<= 3 tiles) that fits 64 KB
    const int Kt = K8 * ntap;
    const int Kp = ((Kt >> 2) & 1) ? Kt : Kt + 4;
    int groups1 = (tiles + 2) / 3;
    int nt1 = (tiles + groups1 - 1) / groups1;
    // small problems (deep U-Net levels): favour many narrow column groups over A re-use - the
    // row panel is L2-resident there and parallelism is what is missing
    if ((long)((a.M + 127) / 128) * groups1 * a.batches < 1024) nt1 = 1;
    if (a.epi == UIR_EPI_PLAIN_DW) nt1 = tiles;                 // the fused Gram belongs to ONE column group
    while (nt1 > 1 && (size_t)nt1 * 32 * Kp * 4 > 60 * 1024) { --nt1; }
    groups1 = (tiles + nt1 - 1) / nt1;
    // (a float4 epilogue through a per-wave LDS transpose was tried and measured slower: profiles/r01/d_ab_notes.md)
    size_t lds = (size_t)nt1 * 32 * Kp * 4 + (size_t)2 * K8 * 4 + (size_t)(nt1 * 32 + K8) * 4;
    if (a.epi == UIR_EPI_GATE_BWD_DW && lds < (size_t)nt1 * 1024 * 4) lds = (size_t)nt1 * 1024 * 4;   // 4-wave dW sum
    if (a.epi == UIR_EPI_PLAIN_DW && lds < (size_t)1024 * 4) lds = (size_t)1024 * 4;
    if (a.epi == UIR_EPI_GELU_BWD && lds < (size_t)4 * nt1 * 32 * 4) lds = (size_t)4 * nt1 * 32 * 4;      // 4-wave bias-gradient sum
    const bool combo_ok = (a.pro == UIR_PRO_NONE) || (a.epi == UIR_EPI_PLAIN && !(a.pro == UIR_PRO_LN && a.R));
    if (lds <= 100 * 1024 && combo_ok) {
      const int rtiles = (a.M + 127) / 128;
      constexpr int gx_total = 768;   // = the resident workgroups at 3 per CU (1536: +0.15 ms/step)
      int gx = gx_total / (groups1 * a.batches);
      if (gx < 1) gx = 1;
      // the variants with fused reductions run 2 workgroups per CU (launch bounds): 512 resident workgroups take all
      // the tiles, and every workgroup writes one slab row for uir_reduce_slab - no point in more rows than that
      constexpr int slab_rows = UIR_GEMM_SLAB_ROWS;
      if ((a.epi == UIR_EPI_GATE_BWD_DW || a.epi == UIR_EPI_GELU_BWD) && gx > slab_rows) gx = slab_rows;
      if (gx > rtiles) gx = rtiles;
      if (a.epi == UIR_EPI_PLAIN_DW) {
        UIR_REQUIRE(groups1 == 1);
        gx = a.dw_rows;
      }
      dim3 grid1(gx, groups1, a.batches);
#define UIR_L1(NTV, PROV, EPIV, RV) \
  hipLaunchKernelGGL((uir_gemm1x1_kernel<NTV, PROV, EPIV, RV>), grid1, dim3(256), lds, st, a, Kp, rtiles)
#define UIR_L1E(NTV, PROV)                                                          \
  do {                                                                              \
    if (a.epi == UIR_EPI_GATE_BWD) UIR_L1(NTV, PROV, UIR_EPI_GATE_BWD, false);      \
    else if (a.epi == UIR_EPI_GATE_BWD_DW) UIR_L1(NTV, PROV, UIR_EPI_GATE_BWD_DW, false); \
    else if (a.epi == UIR_EPI_LEAKY_BWD) UIR_L1(NTV, PROV, UIR_EPI_LEAKY_BWD, false); \
    else if (a.epi == UIR_EPI_GELU_BWD) UIR_L1(NTV, PROV, UIR_EPI_GELU_BWD, false);   \
    else if (a.epi == UIR_EPI_PLAIN_DW) UIR_L1(NTV, PROV, UIR_EPI_PLAIN_DW, false);   \
    else if (a.R) UIR_L1(NTV, PROV, UIR_EPI_PLAIN, true);                           \
    else UIR_L1(NTV, PROV, UIR_EPI_PLAIN, false);                                   \
  } while (0)
#define UIR_L1P(NTV)                                                                \
  do {                                                                              \
    if (a.pro == UIR_PRO_LN) UIR_L1(NTV, UIR_PRO_LN, UIR_EPI_PLAIN, false);         \
    else if (a.pro == UIR_PRO_GATE) { if (a.R) UIR_L1(NTV, UIR_PRO_GATE, UIR_EPI_PLAIN, true); else UIR_L1(NTV, UIR_PRO_GATE, UIR_EPI_PLAIN, false); } \
    else UIR_L1E(NTV, UIR_PRO_NONE);                                                \
  } while (0)
      switch (nt1) {
        case 1: UIR_L1P(1); break;
        case 2: UIR_L1P(2); break;
        default: UIR_L1P(3); break;
      }
#undef UIR_L1P
#undef UIR_L1E
#undef UIR_L1
      return uir_launch_status();
    }
  }
  UIR_REQUIRE(a.ntap <= 0);          // tap ranges exist only on the LDS-panel path above
  UIR_REQUIRE(a.epi != UIR_EPI_GELU_BWD && a.epi != UIR_EPI_PLAIN_DW);   // ... and so does the GELU-backward epilogue (the caller keeps K <= 128)
  const int groups = (tiles + 3) / 4;
  const int nt = (tiles + groups - 1) / groups;
  dim3 grid((a.M + 127) / 128, groups, a.batches), block(256);
  const int vecw = (a.w_sk == 1 && a.taps == 1 && (a.K & 3) == 0 && a.k_grp_pad <= 0 && (a.w_sn & 3) == 0 &&
                    (a.w_bs & 3) == 0 && uir_aligned16(a.W)) ? 1 : 0;
  switch (nt) {
    case 1: hipLaunchKernelGGL(uir_gemm_kernel<1>, grid, block, 0, st, a, vecw); break;
    case 2: hipLaunchKernelGGL(uir_gemm_kernel<2>, grid, block, 0, st, a, vecw); break;
    case 3: hipLaunchKernelGGL(uir_gemm_kernel<3>, grid, block, 0, st, a, vecw); break;
    default: hipLaunchKernelGGL(uir_gemm_kernel<4>, grid, block, 0, st, a, vecw); break;
  }
  return uir_launch_status();
}
