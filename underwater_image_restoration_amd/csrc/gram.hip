// uir_gram: contraction over pixels on the f32 matrix cores -
//   out[z][n][k][tap] = sum_m A[z][m][n] * pro(B[z][nbr(m,tap)][k])
// i.e. every convolution weight-gradient of the path and the MDTA q.k^T Gram matrix.
// MFMA 32x32x2: two pixels per instruction; A's channel on the MFMA row, B's channel on the
// lane, so both operand loads are 128-B contiguous channel runs of one pixel (HWC rows).
// Each workgroup (4 waves) owns one chunk of rows x one (tap, n-group, k-group) tile set, sums
// its 4 waves through LDS and writes ONE partial slab; uir_reduce_slab sums the chunks in a
// fixed order (bitwise reproducible, no float atomics).
#include "uir_common.h"
#include "../../include/uir_kernels.h"

#ifndef UIR_GRAM_U3
#define UIR_GRAM_U3 3   // row pairs per trip of the 3-tile variants: 3 keeps them at 3 waves/SIMD (4: 174 registers -> 2 waves)
#endif

int uir_conv3x3_wgrad(const UirGramArgs& a, void* stream);     // csrc/conv3m.hip; -1 = not its shape
int uir_gram_split(const UirGramArgs& a, void* stream);        // csrc/gramb.hip; -1 = not its case

namespace {

// EXTRAS: also the column sums of squares / column sums (only the q.k^T Gram and the bias gradients ask for them: the
// 3 extra VALU ops per operand element are pure overhead elsewhere - MFMA and VALU time add up on gfx950)
template <int TN, int TK, int PRO, bool EXTRAS>
__global__ __launch_bounds__(256) void uir_gram_kernel(const UirGramArgs a, int gn, int gk) {
  __shared__ float red[TN * TK * 16 * 64];
  __shared__ float redv[4 * 32 * (2 * TN + TK)];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 31, h2 = lane >> 5;
  const int z = blockIdx.z;
  const int ngroups = a.taps * gn * gk;
  const int tg = blockIdx.x % ngroups, chunk = blockIdx.x / ngroups;
  const int tap = tg / (gn * gk);
  const int rem = tg - tap * (gn * gk);
  const int ng = rem / gk, kg = rem - ng * gk;
  const int n0 = ng * 32 * TN, k0 = kg * 32 * TK;
  const float* __restrict__ A = a.A + (long)z * a.a_bs;
  const float* __restrict__ B = a.B + (long)z * a.b_bs;
  const int row_begin = chunk * a.rows_per_chunk;
  const int row_end = min(a.M, row_begin + a.rows_per_chunk);
  int dy = 0, dx = 0;
  if (a.taps > 1) { dy = tap / 3 - 1; dx = tap - (tap / 3) * 3 - 1; }

  bool acol_ok[TN], bcol_ok[TK];
  float lng[TK], lnb[TK];
#pragma unroll
  for (int t = 0; t < TN; ++t) acol_ok[t] = (n0 + 32 * t + r) < a.N;
#pragma unroll
  for (int t = 0; t < TK; ++t) {
    int k = k0 + 32 * t + r;
    bcol_ok[t] = k < a.K;
    lng[t] = 1.f; lnb[t] = 0.f;
    if (PRO == UIR_GRAM_PRO_LN && bcol_ok[t]) { lng[t] = a.ln_g[k]; lnb[t] = a.ln_b[k]; }
  }

  f32x16 acc[TN][TK];
#pragma unroll
  for (int i = 0; i < TN; ++i)
#pragma unroll
    for (int j = 0; j < TK; ++j)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.f;
  float sqa[TN], csa[TN], sqb[TK];
#pragma unroll
  for (int t = 0; t < TN; ++t) { sqa[t] = 0.f; csa[t] = 0.f; }
#pragma unroll
  for (int t = 0; t < TK; ++t) sqb[t] = 0.f;

  // U row-pairs per trip: every load of the trip is issued before the first MFMA consumes one.
  // Loads are branch-free (clamped addresses + select): a predicated load costs a branch and a
  // vmcnt(0) drain in hipcc's codegen, which serialises the whole trip.
  constexpr int U = (TN * TK == 1) ? 8 : (TN * TK == 3 ? UIR_GRAM_U3 : 4);
  int acol[TN], bcol[TK];
#pragma unroll
  for (int t = 0; t < TN; ++t) acol[t] = min(n0 + 32 * t + r, a.N - 1);
#pragma unroll
  for (int t = 0; t < TK; ++t) bcol[t] = min(k0 + 32 * t + r, a.K - 1);
  const int last_row = row_end - 1;
  const float* statz = (PRO == UIR_GRAM_PRO_LN) ? a.stats + (long)z * a.M * 2 : nullptr;
  // One trip = U row pairs per wave.  Two named register panels, loop unrolled x2: the loads of trip t+1 are
  // issued before the MFMAs of trip t (vmcnt retires in issue order, so a rotating-index prefetch would wait for
  // the newest loads; with static panels the wait before compute(P) only covers P's own, older loads).
  // (plain arrays passed by reference: a struct of arrays is not scalarised by hipcc and lands in scratch)
  constexpr int TG = (PRO == UIR_GRAM_PRO_GATE) ? TK : 1, US = (PRO == UIR_GRAM_PRO_LN) ? U : 1;
  int tph = 0, tpw = 0;      // 3x3 taps: image coordinates of the next row this lane loads (one division here, none per row)
  if (a.taps > 1) {
    const int hw = min(row_begin + 2 * wave + h2, last_row) % (a.H * a.Wd);
    tph = hw / a.Wd;
    tpw = hw - tph * a.Wd;
  }
  auto load = [&](float (&Pav)[U][TN], float (&Pbv)[U][TK], float (&Pbg)[U][TG], float2 (&Pst)[US], float (&Pak)[U],
                  float (&Pbk)[U], int base) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int m = base + 8 * u + h2;
      const bool aok = m < row_end;
      bool bok = aok;
      const int mc = min(m, last_row);
      // uniform base + 32-bit per-lane offsets (element counts < 2^30: checked by the launcher): one VALU op per address
      // instead of a 64-bit multiply-add chain - the loop was as busy with address math as with its MFMAs
      const unsigned oa = (unsigned)mc * (unsigned)a.lda;
#pragma unroll
      for (int t = 0; t < TN; ++t) Pav[u][t] = A[oa + (unsigned)acol[t]];
      int mb = mc;
      if (a.taps > 1) {
        // (tph, tpw) = coordinates of row m, advanced by carry: load() is called with bases 8*U apart, in order
        const int hh = tph + dy, ww = tpw + dx;
        const bool in = aok && hh >= 0 && hh < a.H && ww >= 0 && ww < a.Wd;
        bok = in;
        mb = in ? mc + dy * a.Wd + dx : mc;
        tpw += 8;
        while (tpw >= a.Wd) {
          tpw -= a.Wd;
          if (++tph == a.H) tph = 0;
        }
      }
      Pak[u] = aok ? 1.f : 0.f;
      Pbk[u] = bok ? 1.f : 0.f;
      if constexpr (PRO == UIR_GRAM_PRO_LN) Pst[u] = *reinterpret_cast<const float2*>(statz + 2u * (unsigned)mb);
      const unsigned ob = (unsigned)mb * (unsigned)a.ldb;
#pragma unroll
      for (int t = 0; t < TK; ++t) {
        Pbv[u][t] = B[ob + (unsigned)bcol[t]];
        if constexpr (PRO == UIR_GRAM_PRO_GATE) Pbg[u][t] = B[ob + (unsigned)(a.gate_off + bcol[t])];
      }
    }
  };
  auto compute = [&](float (&Pav)[U][TN], float (&Pbv)[U][TK], float (&Pbg)[U][TG], float2 (&Pst)[US], float (&Pak)[U],
                     float (&Pbk)[U]) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      float av[TN], bv[TK];
#pragma unroll
      for (int t = 0; t < TN; ++t) av[t] = (Pak[u] != 0.f && acol_ok[t]) ? Pav[u][t] : 0.f;
#pragma unroll
      for (int t = 0; t < TK; ++t) {
        float v = Pbv[u][t];
        if constexpr (PRO == UIR_GRAM_PRO_LN) v = (v - Pst[u].x) * Pst[u].y * lng[t] + lnb[t];
        else if constexpr (PRO == UIR_GRAM_PRO_GATE) v = uir_gelu(v) * Pbg[u][t];
        bv[t] = (Pbk[u] != 0.f && bcol_ok[t]) ? v : 0.f;
      }
#pragma unroll
      for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TK; ++j) acc[i][j] = uir_mfma(av[i], bv[j], acc[i][j]);
#pragma unroll
      for (int t = 0; t < TN; ++t) { if constexpr (EXTRAS) { sqa[t] += av[t] * av[t]; csa[t] += av[t]; } }
#pragma unroll
      for (int t = 0; t < TK; ++t) { if constexpr (EXTRAS) sqb[t] += bv[t] * bv[t]; }
    }
  };
  {
    float av0[U][TN], bv0[U][TK], bg0[U][TG], ak0[U], bk0[U], av1[U][TN], bv1[U][TK], bg1[U][TG], ak1[U], bk1[U];
    float2 st0[US], st1[US];
    int base = row_begin + 2 * wave;
    if (base < row_end) load(av0, bv0, bg0, st0, ak0, bk0, base);
    // rows past row_end load clamped addresses and contribute exact zeros (mask 0): no tail code
    for (; base < row_end; base += 16 * U) {
      load(av1, bv1, bg1, st1, ak1, bk1, base + 8 * U);
      compute(av0, bv0, bg0, st0, ak0, bk0);
      load(av0, bv0, bg0, st0, ak0, bk0, base + 16 * U);
      compute(av1, bv1, bg1, st1, ak1, bk1);
    }
  }

  // ---- sum the 4 waves through LDS (fixed order), wave 0 stores ----
  for (int w = 0; w < 4; ++w) {
    if (wave == w) {
#pragma unroll
      for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TK; ++j)
#pragma unroll
          for (int q = 0; q < 16; ++q) {
            float* p = &red[((i * TK + j) * 16 + q) * 64 + lane];
            *p = (w == 0) ? acc[i][j][q] : (*p + acc[i][j][q]);
          }
    }
    __syncthreads();
  }
  float* slab = a.slab + (long)chunk * a.slab_stride;
  {
    const int per_thread = TN * TK * 16 * 64 / 256;
    for (int e = 0; e < per_thread; ++e) {
      const int idx = e * 256 + threadIdx.x;
      const int ln = idx & 63;
      const int q = (idx >> 6) & 15;
      const int tile = idx >> 10;
      const int i = tile / TK, j = tile - i * TK;
      const int n = n0 + 32 * i + uir_mfma_row(q, ln >> 5);
      const int k = k0 + 32 * j + (ln & 31);
      if (n < a.N && k < a.K) {
        const int nr = uir_colmap(n, a.n_grp_real, a.n_grp_pad);
        const int kr = uir_colmap(k, a.k_grp_real, a.k_grp_pad);
        if (nr >= 0 && kr >= 0)
          slab[(long)z * a.o_bs + (long)nr * a.o_sn + (long)kr * a.o_sk + (long)tap * a.o_st] = red[idx];
      }
    }
  }
  // ---- column extras: sum of squares of A / B columns, column sums of A ----
  const bool want_a = (a.sq_a_off >= 0 || a.cs_a_off >= 0) && kg == 0 && tap == 0;
  const bool want_b = a.sq_b_off >= 0 && ng == 0 && tap == 0;
  if (want_a || want_b) {
    constexpr int NV = 2 * TN + TK;
#pragma unroll
    for (int t = 0; t < TN; ++t) {
      float s = sqa[t] + __shfl_xor(sqa[t], 32);
      float c = csa[t] + __shfl_xor(csa[t], 32);
      if (h2 == 0) { redv[(wave * NV + t) * 32 + r] = s; redv[(wave * NV + TN + t) * 32 + r] = c; }
    }
#pragma unroll
    for (int t = 0; t < TK; ++t) {
      float s = sqb[t] + __shfl_xor(sqb[t], 32);
      if (h2 == 0) redv[(wave * NV + 2 * TN + t) * 32 + r] = s;
    }
    __syncthreads();
    for (int e = threadIdx.x; e < 32 * NV; e += 256) {
      const int v = e / 32, c = e & 31;
      float s = 0.f;
      for (int w = 0; w < 4; ++w) s += redv[(w * NV + v) * 32 + c];
      if (v < TN) {
        const int n = n0 + 32 * v + c;
        if (want_a && a.sq_a_off >= 0 && n < a.N) slab[a.sq_a_off + (long)z * a.N + n] = s;
      } else if (v < 2 * TN) {
        const int n = n0 + 32 * (v - TN) + c;
        if (want_a && a.cs_a_off >= 0 && n < a.N) slab[a.cs_a_off + (long)z * a.N + n] = s;
      } else {
        const int k = k0 + 32 * (v - 2 * TN) + c;
        if (want_b && k < a.K) slab[a.sq_b_off + (long)z * a.K + k] = s;
      }
    }
  }
}

// 3x3 weight-gradient with N, K <= 32 (the full-resolution convs: embed, down1, outputl, output):
// all 9 taps in one workgroup - the dY row pair is loaded once and multiplied into 9 accumulator
// tiles against the 9 shifted X rows (neighbours in W are adjacent HWC rows: L1 hits), instead of 9
// independent passes over dY and X.
__global__ __launch_bounds__(256) void uir_gram_taps9_kernel(const UirGramArgs a) {
  // blockIdx.y = kernel row ky: this workgroup accumulates taps 3 ky .. 3 ky + 2 (48 accumulator registers instead of
  // 144: at 300 registers per thread the 9-tap version ran one wave per SIMD and kept the other lanes' kernels off its CUs)
  __shared__ float red[3 * 16 * 64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 31, h2 = lane >> 5;
  const int z = blockIdx.z;
  const int chunk = blockIdx.x;
  const int ky = blockIdx.y, dy = ky - 1;
  const float* __restrict__ A = a.A + (long)z * a.a_bs;
  const float* __restrict__ B = a.B + (long)z * a.b_bs;
  const int row_begin = chunk * a.rows_per_chunk;
  const int row_end = min(a.M, row_begin + a.rows_per_chunk);
  const int last_row = row_end - 1;
  const bool aok_c = r < a.N, bok_c = r < a.K;
  const int acol = min(r, a.N - 1), bcol = min(r, a.K - 1);
  f32x16 acc[3];
#pragma unroll
  for (int t = 0; t < 3; ++t)
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[t][q] = 0.f;
  constexpr int U = 4;
  // (ph, pw) of this lane's first pixel by one division; every later pixel is 8 rows further: the coordinates advance by
  // carry (integer divisions per pixel pair were ~2/3 of this kernel's issue slots: 1 M pixels x 3 kernel rows)
  int pw, ph;
  {
    const int m0 = min(row_begin + 2 * wave + h2, last_row);
    const int hw = m0 % (a.H * a.Wd);
    ph = hw / a.Wd;
    pw = hw - ph * a.Wd;
  }
  for (int base = row_begin + 2 * wave; base < row_end; base += 8 * U) {
    float av[U], bv[U][3];
    bool ok[U][3], aok[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int m = base + 8 * u + h2;
      aok[u] = m < row_end;
      const int mc = min(m, last_row);
      av[u] = A[(long)mc * a.lda + acol];
      const int hh = ph + dy;
#pragma unroll
      for (int t = 0; t < 3; ++t) {
        const int dx = t - 1;
        const int ww = pw + dx;
        const bool in = aok[u] && hh >= 0 && hh < a.H && ww >= 0 && ww < a.Wd;
        ok[u][t] = in && bok_c;
        const long mb = in ? (long)mc + dy * a.Wd + dx : (long)mc;
        bv[u][t] = B[mb * a.ldb + bcol];
      }
      pw += 8;
      while (pw >= a.Wd) {
        pw -= a.Wd;
        if (++ph == a.H) ph = 0;
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const float avv = (aok[u] && aok_c) ? av[u] : 0.f;
#pragma unroll
      for (int t = 0; t < 3; ++t) acc[t] = uir_mfma(avv, ok[u][t] ? bv[u][t] : 0.f, acc[t]);
    }
  }
  for (int w = 0; w < 4; ++w) {
    if (wave == w) {
#pragma unroll
      for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          float* p = &red[(t * 16 + q) * 64 + lane];
          *p = (w == 0) ? acc[t][q] : (*p + acc[t][q]);
        }
    }
    __syncthreads();
  }
  float* slab = a.slab + (long)chunk * a.slab_stride;
  for (int idx = threadIdx.x; idx < 3 * 16 * 64; idx += 256) {
    const int ln = idx & 63, q = (idx >> 6) & 15, t = 3 * ky + (idx >> 10);
    const int n = uir_mfma_row(q, ln >> 5), k = ln & 31;
    if (n < a.N && k < a.K) {
      const int nr = uir_colmap(n, a.n_grp_real, a.n_grp_pad);
      const int kr = uir_colmap(k, a.k_grp_real, a.k_grp_pad);
      if (nr >= 0 && kr >= 0) slab[(long)z * a.o_bs + (long)nr * a.o_sn + (long)kr * a.o_sk + (long)t * a.o_st] = red[idx];
    }
  }
}

// 3x3 weight-gradient with few output channels (9*N <= 160, K <= 32: output, outputl, down1, down2 at full / half resolution).
// The kernel above pads N to a 32-row MFMA tile and spends one MFMA per tap; here the roles are swapped so that nothing but the
// last column tile is padding:  dW[n][k][tap] = sum_q X[q][k] * dY[q - off(tap)][n]  (q = px + off), i.e. ONE row operand X^T
// (row = k, exactly the <= 32 input channels) against a column operand whose 9*N columns are the nine shifted copies of dY:
// NT = ceil(9N/32) MFMAs per pixel pair instead of 9 (N = 8: 3), X is read once instead of three times, no kernel-row split.
template <int NT>
__global__ __launch_bounds__(256) void uir_gram_taps9_cols_kernel(const UirGramArgs a) {
  __shared__ float red[NT * 16 * 64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 31, h2 = lane >> 5;
  const int z = blockIdx.z;
  const int chunk = blockIdx.x;
  const float* __restrict__ A = a.A + (long)z * a.a_bs;     // dY [M][lda], N columns
  const float* __restrict__ B = a.B + (long)z * a.b_bs;     // X  [M][ldb], K columns
  const int row_begin = chunk * a.rows_per_chunk;
  const int row_end = min(a.M, row_begin + a.rows_per_chunk);
  const int last_row = row_end - 1;
  const bool kok = r < a.K;
  const int kcol = min(r, a.K - 1);
  // this lane's column of tile t: (tap, n) = divmod(32 t + r, N); the row it reads is q - (dy*Wd + dx)
  int ncol[NT], sdy[NT], sdx[NT], delta[NT];
  bool cok[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int c = 32 * t + r;
    const int tap = min(c / a.N, 8);
    cok[t] = c < 9 * a.N;
    ncol[t] = c - (c / a.N) * a.N;
    sdy[t] = tap / 3 - 1;
    sdx[t] = tap - (tap / 3) * 3 - 1;
    delta[t] = sdy[t] * a.Wd + sdx[t];
  }
  f32x16 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[t][q] = 0.f;
  int pw, ph;                    // coordinates of this lane's next row, advanced by carry (one division here)
  {
    const int m0 = min(row_begin + 2 * wave + h2, last_row);
    const int hw = m0 % (a.H * a.Wd);
    ph = hw / a.Wd;
    pw = hw - ph * a.Wd;
  }
  constexpr int U = 4;
  for (int base = row_begin + 2 * wave; base < row_end; base += 8 * U) {
    float xv[U], yv[U][NT];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int m = base + 8 * u + h2;
      const bool rok = m < row_end;
      const int mc = min(m, last_row);
      const float x = B[(unsigned)mc * (unsigned)a.ldb + (unsigned)kcol];
      xv[u] = (rok && kok) ? x : 0.f;
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const int hh = ph - sdy[t], ww = pw - sdx[t];
        const bool in = rok && cok[t] && hh >= 0 && hh < a.H && ww >= 0 && ww < a.Wd;
        const int ma = in ? mc - delta[t] : mc;
        const float y = A[(unsigned)ma * (unsigned)a.lda + (unsigned)ncol[t]];
        yv[u][t] = in ? y : 0.f;
      }
      pw += 8;
      while (pw >= a.Wd) {
        pw -= a.Wd;
        if (++ph == a.H) ph = 0;
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int t = 0; t < NT; ++t) acc[t] = uir_mfma(xv[u], yv[u][t], acc[t]);
  }
  for (int w = 0; w < 4; ++w) {
    if (wave == w) {
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          float* p = &red[(t * 16 + q) * 64 + lane];
          *p = (w == 0) ? acc[t][q] : (*p + acc[t][q]);
        }
    }
    __syncthreads();
  }
  float* slab = a.slab + (long)chunk * a.slab_stride;
  for (int idx = threadIdx.x; idx < NT * 16 * 64; idx += 256) {
    const int ln = idx & 63, q = (idx >> 6) & 15, t = idx >> 10;
    const int k = uir_mfma_row(q, ln >> 5);          // accumulator row = X channel
    const int c = 32 * t + (ln & 31);                // accumulator column = (tap, n)
    if (k < a.K && c < 9 * a.N) {
      const int tap = c / a.N, n = c - tap * a.N;
      const int nr = uir_colmap(n, a.n_grp_real, a.n_grp_pad);
      const int kr = uir_colmap(k, a.k_grp_real, a.k_grp_pad);
      if (nr >= 0 && kr >= 0) slab[(long)z * a.o_bs + (long)nr * a.o_sn + (long)kr * a.o_sk + (long)tap * a.o_st] = red[idx];
    }
  }
}

// out[i] = sum_c slab[c][i]: 64 outputs per workgroup, the chunks split over 4 waves with 8 independent
// accumulators each (memory-level parallelism), combined in a fixed order.  256 threads: a 1024-thread block needs 16
// free wave slots on ONE CU at once (A/B against the 1024-thread version in the full step: no difference).
__global__ __launch_bounds__(256) void uir_reduce_slab_kernel(const float* __restrict__ slab, int nchunks, long stride,
                                                              float* __restrict__ out, long n, int accumulate, float scale) {
  __shared__ float red[4][64];
  const int lane = threadIdx.x & 63, cl = threadIdx.x >> 6;
  const long i = (long)blockIdx.x * 64 + lane;
  float s[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) s[k] = 0.f;
  if (i < n) {
    const float* p = slab + i;
    int c = cl;
    for (; c + 28 < nchunks; c += 32) {
#pragma unroll
      for (int k = 0; k < 8; ++k) s[k] += p[(long)(c + 4 * k) * stride];
    }
    for (; c < nchunks; c += 4) s[0] += p[(long)c * stride];
  }
  red[cl][lane] = ((s[0] + s[1]) + (s[2] + s[3])) + ((s[4] + s[5]) + (s[6] + s[7]));
  __syncthreads();
  if (cl == 0 && i < n) {
    float t = ((red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane])) * scale;
    out[i] = accumulate ? out[i] + t : t;
  }
}

}  // namespace

extern "C" int uir_gram(const UirGramArgs* p, void* stream) {
  const UirGramArgs& a = *p;
  UIR_REQUIRE(a.A && a.B && a.slab);
  UIR_REQUIRE(a.M > 0 && a.N > 0 && a.K > 0 && a.batches > 0 && a.nchunks > 0);
  UIR_REQUIRE(a.taps == 1 || a.taps == 9);
  UIR_REQUIRE(a.taps == 1 || (a.H > 0 && a.Wd > 0 && a.M % (a.H * a.Wd) == 0));
  UIR_REQUIRE(a.rows_per_chunk > 0 && (a.rows_per_chunk & 7) == 0);
  UIR_REQUIRE((long)a.rows_per_chunk * a.nchunks >= a.M);
  UIR_REQUIRE(a.lda >= a.N && a.ldb >= a.K);
  if (a.pro_b == UIR_GRAM_PRO_LN) UIR_REQUIRE(a.stats && a.ln_g && a.ln_b && a.taps == 1);
  if (a.pro_b == UIR_GRAM_PRO_GATE) UIR_REQUIRE(a.ldb >= a.gate_off + a.K);
  hipStream_t st = static_cast<hipStream_t>(stream);
  dim3 block(256);
#define UIR_GRAM_LAUNCH(TNV, TKV, GN, GK)                                                                              \
  do {                                                                                                                 \
    if (a.pro_b == UIR_GRAM_PRO_LN) hipLaunchKernelGGL((uir_gram_kernel<TNV, TKV, UIR_GRAM_PRO_LN, false>), grid, block, 0, st, a, GN, GK);         \
    else if (a.pro_b == UIR_GRAM_PRO_GATE) hipLaunchKernelGGL((uir_gram_kernel<TNV, TKV, UIR_GRAM_PRO_GATE, false>), grid, block, 0, st, a, GN, GK); \
    else if (extras) hipLaunchKernelGGL((uir_gram_kernel<TNV, TKV, UIR_GRAM_PRO_NONE, true>), grid, block, 0, st, a, GN, GK);                       \
    else hipLaunchKernelGGL((uir_gram_kernel<TNV, TKV, UIR_GRAM_PRO_NONE, false>), grid, block, 0, st, a, GN, GK);                                  \
  } while (0)
  UIR_REQUIRE((long)a.M * a.lda < (1L << 30) && (long)a.M * a.ldb < (1L << 30));   // 32-bit element offsets per batch
  const bool extras = a.sq_a_off >= 0 || a.sq_b_off >= 0 || a.cs_a_off >= 0;
  UIR_REQUIRE(!extras || a.pro_b == UIR_GRAM_PRO_NONE);
  {
    // long pixel contractions with >= 32 columns on both sides and a narrow side (the deep levels' dW_in / dW_qkv / dW_out, the 1x1
    // weight gradients of the up-samplers): LDS-staged split-bf16 MFMA with transposed fragment reads (csrc/gramb.hip)
    const int rs = uir_gram_split(a, stream);
    if (rs >= 0) return rs;
  }
  {
    // Wide weight gradients (Linear layers of NewBigFRFNModel / AST, N and K >= UIR_TILED_GRAM_MIN, default 128): 128 x 128
    // output tiles on the LDS-tiled kernel of csrc/tiled.hip, one k split per chunk - same partial-slab contract.
    constexpr int minw = 128;
    const bool wide = a.N >= minw && a.K >= minw;
    const bool narrow_k = a.N >= minw && a.K == 64, narrow_n = a.N == 64 && a.K >= minw;   // 128 x 64 tiles (the 64 side as second operand)
    if (minw > 0 && (wide || narrow_k || narrow_n) && a.taps == 1 && a.batches == 1 && a.pro_b == UIR_GRAM_PRO_NONE && !extras &&
        a.n_grp_pad <= 0 && a.k_grp_pad <= 0 && ((a.N | a.K | a.lda | a.ldb | a.rows_per_chunk | a.M) & 3) == 0 &&
        uir_aligned16(a.A) && uir_aligned16(a.B)) {
      UirTiledArgs t{};
      if (!narrow_n) {
        t.A = a.A; t.sa_i = 1; t.sa_k = a.lda;
        t.B = a.B; t.sb_j = 1; t.sb_k = a.ldb;
        t.sd_i = a.o_sn; t.sd_j = a.o_sk;
        t.MM = a.N; t.NN = a.K;
      } else {                       // out[n][k] with only 64 rows n: the roles of the operands swap, the output strides with them
        t.A = a.B; t.sa_i = 1; t.sa_k = a.ldb;
        t.B = a.A; t.sb_j = 1; t.sb_k = a.lda;
        t.sd_i = a.o_sk; t.sd_j = a.o_sn;
        t.MM = a.K; t.NN = a.N;
      }
      t.D = a.slab; t.d_split = a.slab_stride;
      t.KK = a.M; t.k_per_split = a.rows_per_chunk;
      t.allow_split = a.N >= 256 || a.K >= 256;
      return uir_tiled_gemm(t, st);
    }
  }
  {
    // 3 x 3 weight gradients with <= 16 outputs and 8 / 16 / 32 inputs: the LDS-staged pixel contraction of csrc/conv3m.hip
    const int rw = uir_conv3x3_wgrad(a, stream);
    if (rw >= 0) return rw;
  }
  const int tn_tiles = (a.N + 31) / 32, tk_tiles = (a.K + 31) / 32;
  constexpr bool cols9 = true;
  if (cols9 && a.taps == 9 && 9 * a.N <= 160 && tk_tiles == 1 && !extras && a.pro_b == UIR_GRAM_PRO_NONE) {
    dim3 grid(a.nchunks, 1, a.batches);
    const int nt = (9 * a.N + 31) / 32;
    if (nt <= 2) hipLaunchKernelGGL(uir_gram_taps9_cols_kernel<2>, grid, block, 0, st, a);
    else if (nt == 3) hipLaunchKernelGGL(uir_gram_taps9_cols_kernel<3>, grid, block, 0, st, a);
    else hipLaunchKernelGGL(uir_gram_taps9_cols_kernel<5>, grid, block, 0, st, a);
  } else if (a.taps == 9 && tn_tiles == 1 && tk_tiles == 1 && a.sq_a_off < 0 && a.sq_b_off < 0 && a.cs_a_off < 0 &&
      a.pro_b == UIR_GRAM_PRO_NONE) {
    dim3 grid(a.nchunks, 3, a.batches);
    hipLaunchKernelGGL(uir_gram_taps9_kernel, grid, block, 0, st, a);
  } else if (tk_tiles == 1 || tn_tiles == 1) {
    // one operand is a single 32-column tile: the other is split into the fewest groups of <= 4 tiles, each group
    // as narrow as possible (6 tiles -> 2 x 3, not 4 + 2 padded: a padded tile still issues its clamped loads)
    const int tiles = tk_tiles == 1 ? tn_tiles : tk_tiles;
    const int groups = (tiles + 3) / 4;
    const int T = (tiles + groups - 1) / groups;
    const int gn = tk_tiles == 1 ? groups : 1, gk = tk_tiles == 1 ? 1 : groups;
    dim3 grid(a.taps * groups * a.nchunks, 1, a.batches);
    if (tk_tiles == 1) {
      switch (T) {
        case 1: UIR_GRAM_LAUNCH(1, 1, gn, gk); break;
        case 2: UIR_GRAM_LAUNCH(2, 1, gn, gk); break;
        case 3: UIR_GRAM_LAUNCH(3, 1, gn, gk); break;
        default: UIR_GRAM_LAUNCH(4, 1, gn, gk); break;
      }
    } else {
      switch (T) {
        case 2: UIR_GRAM_LAUNCH(1, 2, gn, gk); break;
        case 3: UIR_GRAM_LAUNCH(1, 3, gn, gk); break;
        default: UIR_GRAM_LAUNCH(1, 4, gn, gk); break;
      }
    }
  } else {
    const int gn = (tn_tiles + 1) / 2, gk = (tk_tiles + 1) / 2;
    dim3 grid(a.taps * gn * gk * a.nchunks, 1, a.batches);
    UIR_GRAM_LAUNCH(2, 2, gn, gk);
  }
#undef UIR_GRAM_LAUNCH
  return uir_launch_status();
}

// Many slab reductions in ONE launch (the ~380 per-tensor launches of a train step were the #1 symbol of a rocprofv3 trace:
// 7-30 us each for kilobytes of data).  Every workgroup finds its descriptor by binary search over the block prefix table
// and then runs exactly uir_reduce_slab_kernel's summation (same 64-output blocks, same chunk split over the 4 waves, same
// combination order): results are bit-identical to the per-tensor launches.
__global__ __launch_bounds__(256) void uir_reduce_slab_batch_kernel(const UirReduceDesc* __restrict__ descs,
                                                                    const int* __restrict__ first_block, int ndesc) {
  __shared__ float red[4][64];
  int lo = 0, hi = ndesc - 1;
  const int bid = blockIdx.x;
  while (lo < hi) {                       // last d with first_block[d] <= bid
    const int mid = (lo + hi + 1) >> 1;
    if (first_block[mid] <= bid) lo = mid; else hi = mid - 1;
  }
  const UirReduceDesc d = descs[lo];
  const int lane = threadIdx.x & 63, cl = threadIdx.x >> 6;
  const long i = (long)(bid - first_block[lo]) * 64 + lane;
  float s[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) s[k] = 0.f;
  if (i < d.n) {
    const float* p = d.slab + i;
    int c = cl;
    for (; c + 28 < d.nchunks; c += 32) {
#pragma unroll
      for (int k = 0; k < 8; ++k) s[k] += p[(long)(c + 4 * k) * d.stride];
    }
    for (; c < d.nchunks; c += 4) s[0] += p[(long)c * d.stride];
  }
  red[cl][lane] = ((s[0] + s[1]) + (s[2] + s[3])) + ((s[4] + s[5]) + (s[6] + s[7]));
  __syncthreads();
  if (cl == 0 && i < d.n) {
    const float t = ((red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane])) * d.scale;
    d.out[i] = d.accumulate ? d.out[i] + t : t;
  }
}

extern "C" int uir_reduce_slab_batch(const UirReduceDesc* descs_dev, const int* first_block_dev, int ndesc, int total_blocks,
                                     void* stream) {
  UIR_REQUIRE(descs_dev && first_block_dev && ndesc > 0 && total_blocks > 0);
  hipLaunchKernelGGL(uir_reduce_slab_batch_kernel, dim3((unsigned)total_blocks), dim3(256), 0, static_cast<hipStream_t>(stream),
                     descs_dev, first_block_dev, ndesc);
  return uir_launch_status();
}

extern "C" int uir_reduce_slab(const float* slab, int nchunks, long stride, float* out, long n, int accumulate,
                               float scale, void* stream) {
  UIR_REQUIRE(slab && out && nchunks > 0 && n > 0);
  const long blocks = (n + 63) / 64;
  UIR_REQUIRE(blocks < (1L << 31));
  hipLaunchKernelGGL(uir_reduce_slab_kernel, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream),
                     slab, nchunks, stride, out, n, accumulate, scale);
  return uir_launch_status();
}
