// LDS-tiled f32 MFMA contraction for the WIDE layers of the NewBigFRFNModel / AST families: Linear
// layers with 256 ... 2048 input channels (src/model/block.py:223-282, src/Models/AST.py:294-326: LeFF / FRFN linear1 / linear2;
// AST.py:42-64 to_q / to_kv / proj) and their weight gradients.  uir_gemm keeps a whole weight panel of one column group in LDS
// (K <= ~768) and otherwise fell back to an un-staged kernel at 22-29 TFLOP/s; uir_gram's 64 x 64 output groups re-read their
// operands N / 64 resp. K / 64 times.  Here:
//
//     D[i][j] (+ epilogue) = sum_k A(i, k) * B(j, k)          128 x 128 output tile per workgroup, k in chunks of 32
//
// (128 x 64 when the second operand has <= 64 rows) with either operand "k-contiguous" (activation rows [M][K], weights [N][K]) or "index-contiguous" (the token contraction of a
// weight gradient: A(n, m) = dY[m][n]; transposed weights W[k][n] of an input gradient).  Both operand chunks are staged through
// LDS with 16-byte global loads in the operand's own contiguous direction (next chunk in flight during the MFMAs, two LDS stages,
// one barrier per chunk); 4 waves as 2 x 2, each 64 x 64 = four v_mfma_f32_32x32x2_f32 accumulators: 64 MFMAs (4096 cycles) per
// 16 LDS reads per wave and chunk.  Split-k over blockIdx.y writes partial slabs (deterministic; summed by uir_reduce_slab).
#include "uir_common.h"
#include "../../include/uir_kernels.h"
#include <algorithm>
#include <cstdlib>
#include <string>

namespace {

constexpr int TM = 128, TN = 128, TK = 32;
constexpr int KP = TK + 4;        // k-contiguous layout [128][36]: 16-byte rows, stride/4 odd -> conflict-free ds_read_b128
constexpr int IP = TM + 4;        // index-contiguous layout [32][132]
constexpr int STAGE = TM * KP;    // floats per operand per stage (>= 32 * IP)

// global -> registers: the operand's ROWS x 32 chunk as ROWS / 32 float4 per thread (ROWS = 128, or 64 for a narrow B operand)
template <bool KC, int ROWS>
struct Loader {
  static constexpr int NQ = ROWS / 32, IQ = ROWS / 4;
  const float* p[NQ];
  int kk[NQ];                                       // k offset of each float4 inside the chunk (KC: its first k; IC: its k)
  __device__ __forceinline__ void init(const float* base, long s_idx, long s_k, int idx0, int n_idx, int tid) {
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      const int e = tid + 256 * q;
      if (KC) {
        int r = idx0 + (e >> 3);
        r = r < n_idx ? r : n_idx - 1;              // clamped rows are computed and never stored
        kk[q] = 4 * (e & 7);
        p[q] = base + (long)r * s_idx;            // + k at load time
      } else {
        int i = idx0 + 4 * (e % IQ);
        i = i + 4 <= n_idx ? i : n_idx - 4;
        kk[q] = e / IQ;
        p[q] = base + i;
      }
    }
  }
  __device__ __forceinline__ void load(float4 (&v)[NQ], int k0, int ke, long s_k) const {
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      const int k = k0 + kk[q];
      const int kc = k < ke ? k : ke - 4;           // address stays inside the operand; the value is zeroed below
      const float4 t = *reinterpret_cast<const float4*>(KC ? p[q] + kc : p[q] + (long)kc * s_k);
      v[q] = k < ke ? t : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
  __device__ __forceinline__ void store(float* s, const float4 (&v)[NQ], int tid) const {
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      const int e = tid + 256 * q;
      float* d = KC ? s + (e >> 3) * KP + 4 * (e & 7) : s + (e / IQ) * (ROWS + 4) + 4 * (e % IQ);
      *reinterpret_cast<float4*>(d) = v[q];
    }
  }
};

// one operand fragment of a 32-row tile for k sub-block jj: the four k values 8 jj + 4 h2 + e of this lane's row
template <bool KC, int ROWS = TM>
__device__ __forceinline__ float4 frag(const float* s, int row, int jj, int h2) {
  if (KC) return *reinterpret_cast<const float4*>(s + row * KP + 8 * jj + 4 * h2);
  const float* c = s + (8 * jj + 4 * h2) * (ROWS + 4) + row;
  return make_float4(c[0], c[ROWS + 4], c[2 * (ROWS + 4)], c[3 * (ROWS + 4)]);
}

// Row-major epilogue (round 3, measured on the 16-bit kernel first: a 64 -> 256 Linear on 1 M tokens 521 -> 336 us, fill_ on this box
// writes 6.7 TB/s).  An MFMA accumulator holds one output COLUMN per lane, so a store instruction of the plain epilogue is 2 rows x
// 128 bytes of dwords.  Here each 32-row band of the wave's tile goes through the wave's own LDS slice (the operand stages are dead:
// every wave has passed the k loop's last barrier) and leaves as float4: a store instruction is 4 rows x 256 bytes (8 x 128 for a
// 32-column tile), and the bias, the residual and the pre-activation copy move 16 bytes per lane too.  Needs 16-byte addressable rows
// (UirTiledArgs.vec_epi, set by uir_tiled_gemm); per-wave LDS only, no workgroup barrier.
template <int RT, int CT>
__device__ __forceinline__ void uir_rowmajor_epilogue(const UirTiledArgs& t, float* D, float* ws, const f32x16 (&acc)[RT][CT], int row0,
                                                      int col0, int lane) {
  constexpr int EW = 32 * CT + 4;          // floats per LDS row: 16-byte rows, banks shifted by 4 per row
  constexpr int LPR = 8 * CT;              // lanes per row (one float4 each)
  constexpr int RPI = 64 / LPR;            // rows per load / store instruction
  constexpr int NI = 32 / RPI;
  const int li = lane & 31, h2 = lane >> 5;
  const bool gelu = t.act == UIR_ACT_GELU;
  const int c4 = 4 * (lane % LPR), rl = lane / LPR;
  const int col = col0 + c4;
  const bool cok = col < t.NN;             // NN is a multiple of 4: the four columns are in or out together
  const float4 bj = (t.bias && cok) ? *reinterpret_cast<const float4*>(t.bias + col) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int a = 0; a < RT; ++a) {
#pragma unroll
    for (int b = 0; b < CT; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) ws[uir_mfma_row(r, h2) * EW + 32 * b + li] = acc[a][b][r];
    __builtin_amdgcn_wave_barrier();
    const int rb = row0 + 32 * a;
    float sc0 = 1.f, sc1 = 1.f;
    int sc_bnd = 0;
    if (t.out_scale) {                     // per-sample output factor: 32 consecutive rows touch at most two samples (scale_rows >= 32)
      const int s0 = (rb < t.MM ? rb : t.MM - 1) / t.scale_rows;
      sc_bnd = (s0 + 1) * t.scale_rows;
      sc0 = t.out_scale[s0];
      sc1 = t.out_scale[sc_bnd < t.MM ? s0 + 1 : s0];
    }
    float4 rv[NI];                         // the band's residual values in flight together (R may alias D)
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int row = rb + rl + RPI * i;
      rv[i] = (t.R && cok && row < t.MM) ? *reinterpret_cast<const float4*>(t.R + (long)row * t.ldr + col) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int rr = rl + RPI * i, row = rb + rr;
      float4 v = *reinterpret_cast<const float4*>(ws + rr * EW + c4);
      if (cok && row < t.MM) {
        v.x += bj.x; v.y += bj.y; v.z += bj.z; v.w += bj.w;
        if (t.pre_out) *reinterpret_cast<float4*>(t.pre_out + (long)row * t.ldpre + col) = v;
        if (gelu) { v.x = uir_gelu(v.x); v.y = uir_gelu(v.y); v.z = uir_gelu(v.z); v.w = uir_gelu(v.w); }
        if (t.out_scale) { const float sc = row < sc_bnd ? sc0 : sc1; v.x *= sc; v.y *= sc; v.z *= sc; v.w *= sc; }
        v.x += rv[i].x; v.y += rv[i].y; v.z += rv[i].z; v.w += rv[i].w;
        *reinterpret_cast<float4*>(D + (long)row * t.sd_i + col) = v;
      }
    }
    __builtin_amdgcn_wave_barrier();
  }
}

// TNW = 128: four waves as 2 x 2, each 64 x 64.  TNW = 64 (narrow second operand: K = 64 weight gradients, 64-channel outputs): each
// wave 64 x 32.
// TMW = 64 (few row tiles: the deep U-Net levels of the SpectralTransformer, 16 384 rows): 64-row tiles, each wave 32 rows x TNW / 2
// columns, so a 16 384 x 128 output is 256 workgroups instead of 128.
template <bool AKC, bool BKC, int TNW, int TMW = TM>
__global__ __launch_bounds__(256, 2) void uir_tiled_kernel(const UirTiledArgs t) {
  constexpr int CT = TNW / 64;             // 32-column tiles per wave
  constexpr int RT = TMW / 64;             // 32-row tiles per wave
  extern __shared__ float smem[];
  float* As = smem;                       // [2][STAGE]
  float* Bs = smem + 2 * STAGE;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, h2 = lane >> 5;
  const int wr = wave >> 1, wc = wave & 1;
  const int tiles_j = (t.NN + TNW - 1) / TNW;
  const int i0 = (blockIdx.x / tiles_j) * TMW, j0 = (blockIdx.x % tiles_j) * TNW;
  const int split = blockIdx.y;
  const int ks = split * t.k_per_split;
  const int ke = ks + t.k_per_split < t.KK ? ks + t.k_per_split : t.KK;

  Loader<AKC, TMW> la;
  Loader<BKC, TNW> lb;
  la.init(t.A, t.sa_i, t.sa_k, i0, t.MM, tid);
  lb.init(t.B, t.sb_j, t.sb_k, j0, t.NN, tid);

  f32x16 acc[RT][CT];
#pragma unroll
  for (int a = 0; a < RT; ++a)
#pragma unroll
    for (int b = 0; b < CT; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  float4 va[TMW / 32], vb[TNW / 32];
  la.load(va, ks, ke, t.sa_k);
  lb.load(vb, ks, ke, t.sb_k);
  la.store(As, va, tid);
  lb.store(Bs, vb, tid);
  __syncthreads();
  int stage = 0;
  for (int k0 = ks; k0 < ke; k0 += TK) {
    const bool more = k0 + TK < ke;
    if (more) {
      la.load(va, k0 + TK, ke, t.sa_k);
      lb.load(vb, k0 + TK, ke, t.sb_k);
    }
    const float* as = As + stage * STAGE;
    const float* bs = Bs + stage * STAGE;
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
      float av[RT][4];
#pragma unroll
      for (int a = 0; a < RT; ++a) {
        const float4 aa = frag<AKC, TMW>(as, 32 * RT * wr + 32 * a + li, jj, h2);
        av[a][0] = aa.x; av[a][1] = aa.y; av[a][2] = aa.z; av[a][3] = aa.w;
      }
      float bv[CT][4];
#pragma unroll
      for (int b = 0; b < CT; ++b) {
        const float4 bb = frag<BKC, TNW>(bs, 32 * CT * wc + 32 * b + li, jj, h2);
        bv[b][0] = bb.x; bv[b][1] = bb.y; bv[b][2] = bb.z; bv[b][3] = bb.w;
      }
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int a = 0; a < RT; ++a)
#pragma unroll
          for (int b = 0; b < CT; ++b) acc[a][b] = uir_mfma(av[a][e], bv[b][e], acc[a][b]);
    }
    if (more) {
      la.store(As + (stage ^ 1) * STAGE, va, tid);
      lb.store(Bs + (stage ^ 1) * STAGE, vb, tid);
    }
    __syncthreads();
    stage ^= 1;
  }

  float* D = t.D + (long)split * t.d_split;
  if (t.vec_epi) {
    uir_rowmajor_epilogue<RT, CT>(t, D, smem + wave * (32 * (32 * CT + 4)), acc, i0 + 32 * RT * wr, j0 + 32 * CT * wc, lane);
    return;
  }
  const bool gelu = t.act == UIR_ACT_GELU;
  // ALL residual values of the wave's tiles in flight together, before the first store: R may alias D, so the compiler cannot hoist
  // these loads above an earlier element's store by itself - one exposed HBM latency per element (64 per lane) was the cost of a
  // residual (a 65 536 x 512 -> 128 layer: 94 -> 155 us), one per 16-element block still +35 us
  float rv[RT][CT][16];
#pragma unroll
  for (int a = 0; a < RT; ++a)
#pragma unroll
    for (int b = 0; b < CT; ++b) {
      const int col = j0 + 32 * CT * wc + 32 * b + li;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = i0 + 32 * RT * wr + 32 * a + uir_mfma_row(r, h2);
        rv[a][b][r] = (t.R && row < t.MM && col < t.NN) ? t.R[(long)row * t.ldr + col] : 0.f;
      }
    }
#pragma unroll
  for (int a = 0; a < RT; ++a) {
    // per-sample output factor: 32 consecutive rows touch at most two samples (scale_rows >= 32)
    float sc0 = 1.f, sc1 = 1.f;
    int sc_bnd = 0;
    if (t.out_scale) {
      const int rb = i0 + 32 * RT * wr + 32 * a;
      const int s0 = (rb < t.MM ? rb : t.MM - 1) / t.scale_rows;
      sc_bnd = (s0 + 1) * t.scale_rows;
      sc0 = t.out_scale[s0];
      sc1 = t.out_scale[sc_bnd < t.MM ? s0 + 1 : s0];
    }
#pragma unroll
    for (int b = 0; b < CT; ++b) {
      const int col = j0 + 32 * CT * wc + 32 * b + li;
      const float bj = (t.bias && col < t.NN) ? t.bias[col] : 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = i0 + 32 * RT * wr + 32 * a + uir_mfma_row(r, h2);
        if (row < t.MM && col < t.NN) {
          float v = acc[a][b][r] + bj;
          if (t.pre_out) t.pre_out[(long)row * t.ldpre + col] = v;
          if (gelu) v = uir_gelu(v);
          if (t.out_scale) v *= row < sc_bnd ? sc0 : sc1;
          D[(long)row * t.sd_i + (long)col * t.sd_j] = v + rv[a][b][r];
        }
      }
    }
  }
}


// ------------------------------------------------------------------------------------------------------------------------
// Dense 3 x 3 convolution with MANY input channels on the same tiles (Conv2d of the samplers: AST.py:408-424 after the re-layout of
// Models/AST.py, src/model/block.py:107-153; the SpectralTransformer's UpSample1, SpectralTransformer.py:191-198):
//     Y[m][n] = sum_tap sum_k X[nbr(m, tap)][k] * W[n * sb_j + k * sb_k + wt(tap) * sb_t]        (uir_gemm's taps = 9 contract)
// A chunks are rows of the neighbour pixel (zero outside the image), B chunks are read with scalar loads at the weight's own strides
// (native [N][K][3][3]: k stride 9 - L2-resident, 16 loads per thread and chunk) into the k-contiguous LDS layout.  The k loop runs
// over 9 taps x K / 32 chunks.  uir_gemm kept a whole nine-tap panel in LDS (K <= 80), split the taps over 3-9 launches (K <= 380)
// or fell back to an un-staged kernel (22-29 TFLOP/s) beyond.
struct ConvLoaderA {
  const float* p[4];
  int kk[4], ph[4], pw[4];
  __device__ __forceinline__ void init(const UirTiledArgs& t, int i0, int tid) {
    const int HW = t.H * t.Wd;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int e = tid + 256 * q;
      int r = i0 + (e >> 3);
      r = r < t.MM ? r : t.MM - 1;
      const int hw = r % HW;
      ph[q] = hw / t.Wd;
      pw[q] = hw - ph[q] * t.Wd;
      kk[q] = 4 * (e & 7);
      p[q] = t.A + (long)r * t.sa_i;
    }
  }
  __device__ __forceinline__ void load(float4 (&v)[4], const UirTiledArgs& t, int tap, int k0) const {
    const int dy = tap / 3 - 1, dx = tap - (tap / 3) * 3 - 1;
    const long shift = ((long)dy * t.Wd + dx) * t.sa_i;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int k = k0 + kk[q];
      const int hh = ph[q] + dy, ww = pw[q] + dx;
      const bool ok = k < t.KK && hh >= 0 && hh < t.H && ww >= 0 && ww < t.Wd;
      const float4 x = *reinterpret_cast<const float4*>(p[q] + (ok ? shift + k : 0));     // clamped address, value selected below
      v[q] = ok ? x : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
  __device__ __forceinline__ void store(float* s, const float4 (&v)[4], int tid) const {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int e = tid + 256 * q;
      *reinterpret_cast<float4*>(s + (e >> 3) * KP + 4 * (e & 7)) = v[q];
    }
  }
};

struct ConvLoaderB {                          // element (j, k) of the chunk: j = (tid >> 5) + 8 q, k = tid & 31
  const float* p0;
  long jstep;
  int jbase, k;
  __device__ __forceinline__ void init(const UirTiledArgs& t, int j0, int tid) {
    jbase = j0 + (tid >> 5);
    k = tid & 31;
    p0 = t.B;
    jstep = t.sb_j;
  }
  __device__ __forceinline__ void load(float (&v)[16], const UirTiledArgs& t, int wt, int k0) const {
    const int kg = k0 + k;
    const bool kok = kg < t.KK;
    const long koff = (long)(kok ? kg : 0) * t.sb_k + (long)wt * t.sb_t;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int j = jbase + 8 * q;
      const bool ok = kok && j < t.NN;
      const float x = p0[(long)(j < t.NN ? j : t.NN - 1) * jstep + koff];
      v[q] = ok ? x : 0.f;
    }
  }
  __device__ __forceinline__ void store(float* s, const float (&v)[16], int tid) const {
#pragma unroll
    for (int q = 0; q < 16; ++q) s[((tid >> 5) + 8 * q) * KP + (tid & 31)] = v[q];
  }
};

__global__ __launch_bounds__(256, 2) void uir_tiled_conv_kernel(const UirTiledArgs t) {
  extern __shared__ float smem[];
  float* As = smem;                       // [2][STAGE]
  float* Bs = smem + 2 * STAGE;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, h2 = lane >> 5;
  const int wr = wave >> 1, wc = wave & 1;
  const int tiles_j = (t.NN + TN - 1) / TN;
  const int i0 = (blockIdx.x / tiles_j) * TM, j0 = (blockIdx.x % tiles_j) * TN;
  const int kchunks = (t.KK + TK - 1) / TK, nchunks = 9 * kchunks;

  ConvLoaderA la;
  ConvLoaderB lb;
  la.init(t, i0, tid);
  lb.init(t, j0, tid);

  f32x16 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  float4 va[4];
  float vb[16];
  la.load(va, t, 0, 0);
  lb.load(vb, t, t.flip ? 8 : 0, 0);
  la.store(As, va, tid);
  lb.store(Bs, vb, tid);
  __syncthreads();
  int stage = 0, tap = 0, kc = 0;           // the chunk being computed
  for (int c = 0; c < nchunks; ++c) {
    int ntap = tap, nkc = kc + 1;            // the next chunk
    if (nkc == kchunks) { nkc = 0; ++ntap; }
    const bool more = c + 1 < nchunks;
    if (more) {
      la.load(va, t, ntap, nkc * TK);
      lb.load(vb, t, t.flip ? 8 - ntap : ntap, nkc * TK);
    }
    const float* as = As + stage * STAGE;
    const float* bs = Bs + stage * STAGE;
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
      const float4 a0 = frag<true>(as, 64 * wr + li, jj, h2), a1 = frag<true>(as, 64 * wr + 32 + li, jj, h2);
      const float4 b0 = frag<true>(bs, 64 * wc + li, jj, h2), b1 = frag<true>(bs, 64 * wc + 32 + li, jj, h2);
      const float av[2][4] = {{a0.x, a0.y, a0.z, a0.w}, {a1.x, a1.y, a1.z, a1.w}};
      const float bv[2][4] = {{b0.x, b0.y, b0.z, b0.w}, {b1.x, b1.y, b1.z, b1.w}};
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
          for (int b = 0; b < 2; ++b) acc[a][b] = uir_mfma(av[a][e], bv[b][e], acc[a][b]);
    }
    if (more) {
      la.store(As + (stage ^ 1) * STAGE, va, tid);
      lb.store(Bs + (stage ^ 1) * STAGE, vb, tid);
    }
    __syncthreads();
    stage ^= 1;
    tap = ntap;
    kc = nkc;
  }

#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const int col = j0 + 64 * wc + 32 * b + li;
      const float bj = (t.bias && col < t.NN) ? t.bias[col] : 0.f;
      float rv[16];                       // residual values in flight together (see uir_tiled_kernel)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = i0 + 64 * wr + 32 * a + uir_mfma_row(r, h2);
        rv[r] = (t.R && row < t.MM && col < t.NN) ? t.R[(long)row * t.ldr + col] : 0.f;
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = i0 + 64 * wr + 32 * a + uir_mfma_row(r, h2);
        if (row < t.MM && col < t.NN) t.D[(long)row * t.sd_i + (long)col * t.sd_j] = acc[a][b][r] + bj + rv[r];
      }
    }
}


// ------------------------------------------------------------------------------------------------------------------------
// The same contraction on the BF16 matrix pipe with SPLIT f32 operands (opt-in: UIR_TILED_SPLIT=1).  On gfx950 the f32 MFMA runs at
// 1/16 of the bf16 MFMA rate (64 vs 1024 FLOP/clk/SIMD).  An f32 value is the exact sum of three bf16 pieces (8 significant bits each:
// hi = top 16 bits of x, mid = top 16 bits of x - hi, lo = top 16 bits of x - hi - mid; the subtractions are exact), a bf16 x bf16 product
// is exact in f32, so   a b = sum over the six piece pairs (h,h) (h,m) (m,h) (h,l) (l,h) (m,m)  up to 3 * 2^-24 |a b| (the dropped m l, l m,
// l l terms) - the rounding level of an f32 multiply.  Six v_mfma_f32_32x32x16_bf16 (6 * 32 cycles for k = 16) replace eight
// v_mfma_f32_32x32x2_f32 (8 * 64 cycles); accumulation stays in f32.  The split is done ONCE per element, when the chunk is staged:
// LDS holds three bf16 planes per operand, [row][k] in 64-byte rows with an XOR swizzle of the 16-byte chunks (conflict-free
// ds_read_b128 of a lane's 8 consecutive k; 48 KB per workgroup, three workgroups = three waves per SIMD in different phases).
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
constexpr int SROW = 32;                   // bf16 per LDS row: 64 bytes = four 16-byte chunks, unpadded; chunk c of row r sits at chunk
                                           // c ^ ((r >> 1) & 3): eight consecutive rows of one logical chunk cover all eight 16-byte bank groups
// and rows r, r + 1 swap places where bit 2 of r is set, so the rows 4 i + e written by consecutive lanes of the index-contiguous loader
                                           // alternate between the two 64-byte halves of a bank line
__device__ __forceinline__ int swz(int row, int chunk) { return (row ^ ((row >> 2) & 1)) * SROW + ((chunk ^ ((row >> 1) & 3)) << 3); }
constexpr int SPLANE = TM * SROW;          // bf16 per plane
constexpr int SSTAGE = 3 * SPLANE;         // bf16 per operand per stage

__device__ __forceinline__ void split3(float x, uint32_t& h, uint32_t& m, uint32_t& l) {
  h = __float_as_uint(x) & 0xffff0000u;
  const float r1 = x - __uint_as_float(h);
  m = __float_as_uint(r1) & 0xffff0000u;
  const float r2 = r1 - __uint_as_float(m);
  l = __float_as_uint(r2) & 0xffff0000u;
}
__device__ __forceinline__ uint32_t pack2(uint32_t lo_elem, uint32_t hi_elem) { return (hi_elem & 0xffff0000u) | (lo_elem >> 16); }

// four consecutive k of one row -> 8 bytes in each of the three planes
__device__ __forceinline__ void store_row4(unsigned short* s, int row, int k, const float (&x)[4]) {
  uint32_t h[4], m[4], l[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) split3(x[e], h[e], m[e], l[e]);
  unsigned short* d = s + swz(row, k >> 3) + (k & 7);
  *reinterpret_cast<uint2*>(d) = make_uint2(pack2(h[0], h[1]), pack2(h[2], h[3]));
  *reinterpret_cast<uint2*>(d + SPLANE) = make_uint2(pack2(m[0], m[1]), pack2(m[2], m[3]));
  *reinterpret_cast<uint2*>(d + 2 * SPLANE) = make_uint2(pack2(l[0], l[1]), pack2(l[2], l[3]));
}

// global -> registers -> split -> LDS for one operand chunk of 128 x 32.
// KC (k-contiguous): 4 float4 per thread, each 4 consecutive k of one row.  IC (index-contiguous): one 4 (k) x 4 (index) block per
// thread = 4 float4 along the index at 4 consecutive k, transposed in registers.
template <bool KC>
struct SplitLoader {
  const float* p[4];
  int kk[4], row0;
  __device__ __forceinline__ void init(const float* base, long s_idx, int idx0, int n_idx, int tid) {
    if (KC) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int e = tid + 256 * q;
        int r = idx0 + (e >> 3);
        r = r < n_idx ? r : n_idx - 1;
        kk[q] = 4 * (e & 7);
        p[q] = base + (long)r * s_idx;
      }
    } else {
      // lanes: k block = tid & 7 (fastest), index quad = tid >> 3: a wave's load instruction covers 8 rows x 128 contiguous bytes,
      // its LDS stores 8 lanes x 8 bytes = one whole 64-byte row half
      int i = idx0 + 4 * (tid >> 3);
      row0 = 4 * (tid >> 3);
      i = i + 4 <= n_idx ? i : n_idx - 4;
#pragma unroll
      for (int q = 0; q < 4; ++q) { kk[q] = 4 * (tid & 7) + q; p[q] = base + i; }
    }
  }
  // RAW load of the chunk at k0 (clamped addresses, no tail mask, unconditional): as first written - a select per float4 under
  // `if (more)` - hipcc put every load under its own exec branch with an s_waitcnt vmcnt(0) behind it: eight serialised round trips
  // per chunk.  The k tail is masked when the chunk is split, after the MFMAs of the chunk before it.
  __device__ __forceinline__ void load_raw(uint4 (&v)[4], int k0, int ke, long s_k) const {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int k = k0 + kk[q];
      const int kc = k < ke ? k : ke - 4;
      v[q] = *reinterpret_cast<const uint4*>(KC ? p[q] + kc : p[q] + (long)kc * s_k);
    }
  }
  __device__ __forceinline__ void store(unsigned short* s, const uint4 (&v)[4], int tid, int k0, int ke) const {
#define UIR_F(x, q) __uint_as_float((x) & m##q)
    const uint32_t m0 = k0 + kk[0] < ke ? 0xffffffffu : 0u, m1 = k0 + kk[1] < ke ? 0xffffffffu : 0u;
    const uint32_t m2 = k0 + kk[2] < ke ? 0xffffffffu : 0u, m3 = k0 + kk[3] < ke ? 0xffffffffu : 0u;
    if (KC) {
      const float x0[4] = {UIR_F(v[0].x, 0), UIR_F(v[0].y, 0), UIR_F(v[0].z, 0), UIR_F(v[0].w, 0)};
      const float x1[4] = {UIR_F(v[1].x, 1), UIR_F(v[1].y, 1), UIR_F(v[1].z, 1), UIR_F(v[1].w, 1)};
      const float x2[4] = {UIR_F(v[2].x, 2), UIR_F(v[2].y, 2), UIR_F(v[2].z, 2), UIR_F(v[2].w, 2)};
      const float x3[4] = {UIR_F(v[3].x, 3), UIR_F(v[3].y, 3), UIR_F(v[3].z, 3), UIR_F(v[3].w, 3)};
      store_row4(s, tid >> 3, 4 * (tid & 7), x0);
      store_row4(s, (tid + 256) >> 3, 4 * (tid & 7), x1);
      store_row4(s, (tid + 512) >> 3, 4 * (tid & 7), x2);
      store_row4(s, (tid + 768) >> 3, 4 * (tid & 7), x3);
    } else {
      const int k = 4 * (tid & 7);
      const float r0[4] = {UIR_F(v[0].x, 0), UIR_F(v[1].x, 1), UIR_F(v[2].x, 2), UIR_F(v[3].x, 3)};
      const float r1[4] = {UIR_F(v[0].y, 0), UIR_F(v[1].y, 1), UIR_F(v[2].y, 2), UIR_F(v[3].y, 3)};
      const float r2[4] = {UIR_F(v[0].z, 0), UIR_F(v[1].z, 1), UIR_F(v[2].z, 2), UIR_F(v[3].z, 3)};
      const float r3[4] = {UIR_F(v[0].w, 0), UIR_F(v[1].w, 1), UIR_F(v[2].w, 2), UIR_F(v[3].w, 3)};
      store_row4(s, row0, k, r0);
      store_row4(s, row0 + 1, k, r1);
      store_row4(s, row0 + 2, k, r2);
      store_row4(s, row0 + 3, k, r3);
    }
#undef UIR_F
  }
};

__device__ __forceinline__ bf16x8 sfrag(const unsigned short* s, int row, int ks, int h2) {
  return __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(s + swz(row, 2 * ks + h2)));
}

template <bool AKC, bool BKC>
__global__ __launch_bounds__(256, 3) void uir_tiled_split_kernel(const UirTiledArgs t) {
  extern __shared__ float smem[];
  unsigned short* As = reinterpret_cast<unsigned short*>(smem);       // [SSTAGE]: one stage (3 planes x 2 operands = 48 KB, 3 WGs / CU);
  unsigned short* Bs = As + SSTAGE;                                   // the next chunk waits in registers, the other workgroup of the CU
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, h2 = lane >> 5;   // covers the two barriers per chunk
  const int wr = wave >> 1, wc = wave & 1;
  const int tiles_j = (t.NN + TN - 1) / TN;
  const int i0 = (blockIdx.x / tiles_j) * TM, j0 = (blockIdx.x % tiles_j) * TN;
  const int split = blockIdx.y;
  const int ks = split * t.k_per_split;
  const int ke = ks + t.k_per_split < t.KK ? ks + t.k_per_split : t.KK;

  SplitLoader<AKC> la;
  SplitLoader<BKC> lb;
  la.init(t.A, t.sa_i, i0, t.MM, tid);
  lb.init(t.B, t.sb_j, j0, t.NN, tid);

  f32x16 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  uint4 va[4], vb[4];
  la.load_raw(va, ks, ke, t.sa_k);
  lb.load_raw(vb, ks, ke, t.sb_k);
  la.store(As, va, tid, ks, ke);
  lb.store(Bs, vb, tid, ks, ke);
  __syncthreads();
  for (int k0 = ks; k0 < ke; k0 += TK) {
    const bool more = k0 + TK < ke;
    la.load_raw(va, k0 + TK, ke, t.sa_k);           // unconditional: past the end the addresses clamp and the values are never stored
    lb.load_raw(vb, k0 + TK, ke, t.sb_k);
    __builtin_amdgcn_sched_barrier(0);              // loads issued; their consumers (mask, split, LDS stores) stay behind the MFMAs
#pragma unroll
    for (int kq = 0; kq < 2; ++kq) {
      bf16x8 fa[2][3], fb[2][3];
#pragma unroll
      for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) {
          fa[x][pl] = sfrag(As + pl * SPLANE, 64 * wr + 32 * x + li, kq, h2);
          fb[x][pl] = sfrag(Bs + pl * SPLANE, 64 * wc + 32 * x + li, kq, h2);
        }
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
          f32x16 c = acc[a][b];
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[a][1], fb[b][1], c, 0, 0, 0);     // small terms first
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[a][2], fb[b][0], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[a][0], fb[b][2], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[a][1], fb[b][0], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[a][0], fb[b][1], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[a][0], fb[b][0], c, 0, 0, 0);
          acc[a][b] = c;
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();                 // every wave has read this chunk
    if (more) {
      la.store(As, va, tid, k0 + TK, ke);
      lb.store(Bs, vb, tid, k0 + TK, ke);
    }
    __syncthreads();
  }

  float* D = t.D + (long)split * t.d_split;
  if (t.vec_epi) {
    uir_rowmajor_epilogue<2, 2>(t, D, smem + wave * (32 * 68), acc, i0 + 64 * wr, j0 + 64 * wc, lane);
    return;
  }
  const bool gelu = t.act == UIR_ACT_GELU;
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const int col = j0 + 64 * wc + 32 * b + li;
      const float bj = (t.bias && col < t.NN) ? t.bias[col] : 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = i0 + 64 * wr + 32 * a + uir_mfma_row(r, h2);
        if (row < t.MM && col < t.NN) {
          float v = acc[a][b][r] + bj;
          if (t.pre_out) t.pre_out[(long)row * t.ldpre + col] = v;
          if (gelu) v = uir_gelu(v);
          if (t.R) v += t.R[(long)row * t.ldr + col];
          D[(long)row * t.sd_i + (long)col * t.sd_j] = v;
        }
      }
    }
}


// ------------------------------------------------------------------------------------------------------------------------
// MIXED precision (opt-in: UIR_MIXED=bf16 | fp16; BASELINE configs[4] names "mixed fp16" for AST): the same contraction with each f32
// operand rounded ONCE (round-to-nearest-even, when its chunk is staged) to bf16 resp. fp16, one v_mfma_f32_32x32x16_{bf16,f16} per
// 16 k, f32 accumulation and the f32 epilogue of uir_tiled_kernel (bias, pre-activation copy, GELU, per-sample factor, residual).
// What torch.autocast does to a Linear - low-precision operands, f32 accumulators, f32 master weights and outputs - and nothing else:
// LayerNorm, softmax, the depthwise convolutions, the losses, the gradient norm and Adam stay f32.  One plane per operand: 8 KB per
// operand and stage, two stages (32 KB per workgroup, one barrier per chunk); LDS rows and fragments as in the split kernel above.
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
constexpr int EPW = 68;                    // floats per LDS row of the row-major epilogue (64 + 4: 16-byte rows, shifted banks)
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

template <bool F16>
__device__ __forceinline__ uint32_t lp_pack2(float a, float b) {       // round-to-nearest-even; v_cvt_pk_{bf16,f16}_f32 on gfx950
  if (F16) {                                                            // saturating: a finite f32 never becomes an infinity
    a = a != a ? a : __builtin_amdgcn_fmed3f(a, -65504.f, 65504.f);     // (a NaN stays a NaN: a diverged run must still show)
    b = b != b ? b : __builtin_amdgcn_fmed3f(b, -65504.f, 65504.f);
    const _Float16 lo = static_cast<_Float16>(a), hi = static_cast<_Float16>(b);
    return (uint32_t)__builtin_bit_cast(unsigned short, lo) | ((uint32_t)__builtin_bit_cast(unsigned short, hi) << 16);
  }
  const __bf16 lo = static_cast<__bf16>(a), hi = static_cast<__bf16>(b);
  return (uint32_t)__builtin_bit_cast(unsigned short, lo) | ((uint32_t)__builtin_bit_cast(unsigned short, hi) << 16);
}
template <bool F16>
__device__ __forceinline__ void lp_store_row4(unsigned short* s, int row, int k, float x0, float x1, float x2, float x3) {
  *reinterpret_cast<uint2*>(s + swz(row, k >> 3) + (k & 7)) = make_uint2(lp_pack2<F16>(x0, x1), lp_pack2<F16>(x2, x3));
}
// SplitLoader's addressing; one rounded plane instead of three truncated ones.  The loads are RAW (clamped addresses); the k-tail
// mask is applied when the chunk is converted, AFTER the MFMAs of the chunk before it (the kernel pins that order with
// sched_barriers): nothing waits on the loads before the matrix work they are meant to hide behind.
template <bool KC, bool F16>
struct LpLoader : SplitLoader<KC> {
  __device__ __forceinline__ void load_lp(uint4 (&v)[4], int k0, int ke, long s_k) const { this->load_raw(v, k0, ke, s_k); }
  __device__ __forceinline__ void store_lp(unsigned short* s, const uint4 (&v)[4], int tid, int k0, int ke) const {
#define UIR_F(x, q) __uint_as_float((x) & m##q)
    if (KC) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int e = tid + 256 * q;
        const uint32_t mq = k0 + this->kk[q] < ke ? 0xffffffffu : 0u;
        lp_store_row4<F16>(s, e >> 3, 4 * (e & 7), __uint_as_float(v[q].x & mq), __uint_as_float(v[q].y & mq),
                           __uint_as_float(v[q].z & mq), __uint_as_float(v[q].w & mq));
      }
    } else {
      const int k = 4 * (tid & 7);
      const uint32_t m0 = k0 + this->kk[0] < ke ? 0xffffffffu : 0u, m1 = k0 + this->kk[1] < ke ? 0xffffffffu : 0u;
      const uint32_t m2 = k0 + this->kk[2] < ke ? 0xffffffffu : 0u, m3 = k0 + this->kk[3] < ke ? 0xffffffffu : 0u;
      lp_store_row4<F16>(s, this->row0, k, UIR_F(v[0].x, 0), UIR_F(v[1].x, 1), UIR_F(v[2].x, 2), UIR_F(v[3].x, 3));
      lp_store_row4<F16>(s, this->row0 + 1, k, UIR_F(v[0].y, 0), UIR_F(v[1].y, 1), UIR_F(v[2].y, 2), UIR_F(v[3].y, 3));
      lp_store_row4<F16>(s, this->row0 + 2, k, UIR_F(v[0].z, 0), UIR_F(v[1].z, 1), UIR_F(v[2].z, 2), UIR_F(v[3].z, 3));
      lp_store_row4<F16>(s, this->row0 + 3, k, UIR_F(v[0].w, 0), UIR_F(v[1].w, 1), UIR_F(v[2].w, 2), UIR_F(v[3].w, 3));
    }
#undef UIR_F
  }
};

template <bool F16>
__device__ __forceinline__ f32x16 lp_mfma(uint4 a, uint4 b, f32x16 c) {
  if (F16) return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

template <bool AKC, bool BKC, bool F16>
__global__ __launch_bounds__(256, 3) void uir_tiled_lp_kernel(const UirTiledArgs t) {
  extern __shared__ float smem[];
  unsigned short* As = reinterpret_cast<unsigned short*>(smem);       // [2][SPLANE]
  unsigned short* Bs = As + 2 * SPLANE;                               // [2][SPLANE]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, h2 = lane >> 5;
  const int wr = wave >> 1, wc = wave & 1;
  const int tiles_j = (t.NN + TN - 1) / TN;
  const int i0 = (blockIdx.x / tiles_j) * TM, j0 = (blockIdx.x % tiles_j) * TN;
  const int split = blockIdx.y;
  const int ks = split * t.k_per_split;
  const int ke = ks + t.k_per_split < t.KK ? ks + t.k_per_split : t.KK;

  LpLoader<AKC, F16> la;
  LpLoader<BKC, F16> lb;
  la.init(t.A, t.sa_i, i0, t.MM, tid);
  lb.init(t.B, t.sb_j, j0, t.NN, tid);

  f32x16 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  uint4 va[4], vb[4];
  la.load_lp(va, ks, ke, t.sa_k);
  lb.load_lp(vb, ks, ke, t.sb_k);
  la.store_lp(As, va, tid, ks, ke);
  lb.store_lp(Bs, vb, tid, ks, ke);
  __syncthreads();
  int stage = 0;
  for (int k0 = ks; k0 < ke; k0 += TK) {
    const bool more = k0 + TK < ke;
    // unconditional (past the end the addresses clamp into the operand and the values are never stored): a load under `if (more)`
    // reaches the stores through loop-carried copies, and hipcc waits for it right there to make them
    la.load_lp(va, k0 + TK, ke, t.sa_k);
    lb.load_lp(vb, k0 + TK, ke, t.sb_k);
    __builtin_amdgcn_sched_barrier(0);              // the loads are issued; nothing of their consumers may move up here
    const unsigned short* as = As + stage * SPLANE;
    const unsigned short* bs = Bs + stage * SPLANE;
#pragma unroll
    for (int kq = 0; kq < 2; ++kq) {
      uint4 fa[2], fb[2];
#pragma unroll
      for (int x = 0; x < 2; ++x) {
        fa[x] = *reinterpret_cast<const uint4*>(as + swz(64 * wr + 32 * x + li, 2 * kq + h2));
        fb[x] = *reinterpret_cast<const uint4*>(bs + swz(64 * wc + 32 * x + li, 2 * kq + h2));
      }
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) acc[a][b] = lp_mfma<F16>(fa[a], fb[b], acc[a][b]);
    }
    __builtin_amdgcn_sched_barrier(0);              // mask + convert + LDS stores of the next chunk stay behind the MFMAs
    if (more) {
      la.store_lp(As + (stage ^ 1) * SPLANE, va, tid, k0 + TK, ke);
      lb.store_lp(Bs + (stage ^ 1) * SPLANE, vb, tid, k0 + TK, ke);
    }
    __syncthreads();                 // every wave has read this stage and written the other one
    stage ^= 1;
  }

  float* D = t.D + (long)split * t.d_split;
  const bool gelu = t.act == UIR_ACT_GELU;
  if (t.vec_epi) {
    uir_rowmajor_epilogue<2, 2>(t, D, smem + wave * (32 * EPW), acc, i0 + 64 * wr, j0 + 64 * wc, lane);
    return;
  }
  // uir_tiled_kernel's epilogue (RT = CT = 2), the residual values of one 32-row band of the wave in flight together before that
  // band's stores (R may alias D; all 64 at once do not fit the 168 registers of three workgroups per CU)
#pragma unroll
  for (int a = 0; a < 2; ++a) {
    float rv[2][16];
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const int col = j0 + 64 * wc + 32 * b + li;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = i0 + 64 * wr + 32 * a + uir_mfma_row(r, h2);
        rv[b][r] = (t.R && row < t.MM && col < t.NN) ? t.R[(long)row * t.ldr + col] : 0.f;
      }
    }
    float sc0 = 1.f, sc1 = 1.f;
    int sc_bnd = 0;
    if (t.out_scale) {
      const int rb = i0 + 64 * wr + 32 * a;
      const int s0 = (rb < t.MM ? rb : t.MM - 1) / t.scale_rows;
      sc_bnd = (s0 + 1) * t.scale_rows;
      sc0 = t.out_scale[s0];
      sc1 = t.out_scale[sc_bnd < t.MM ? s0 + 1 : s0];
    }
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const int col = j0 + 64 * wc + 32 * b + li;
      const float bj = (t.bias && col < t.NN) ? t.bias[col] : 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = i0 + 64 * wr + 32 * a + uir_mfma_row(r, h2);
        if (row < t.MM && col < t.NN) {
          float v = acc[a][b][r] + bj;
          if (t.pre_out) t.pre_out[(long)row * t.ldpre + col] = v;
          if (gelu) v = uir_gelu(v);
          if (t.out_scale) v *= row < sc_bnd ? sc0 : sc1;
          D[(long)row * t.sd_i + (long)col * t.sd_j] = v + rv[b][r];
        }
      }
    }
  }
}

}  // namespace

// 0: off; 1: bf16; 2: fp16; -1: an unknown value (refused by uir_tiled_gemm instead of silently training in f32)
int uir_mixed_mode() {
  static const int mode = [] {
    const char* v = getenv("UIR_MIXED");
    if (!v || !*v || (v[0] == '0' && !v[1])) return 0;
    const std::string s(v);
    return s == "bf16" ? 1 : s == "fp16" ? 2 : -1;
  }();
  return mode;
}

// UIR_FUSED_SPLIT (knobs.py): 1 / unset = the tile-fused kernels (csrc/gdfn.hip, csrc/dwln.hip) contract on split-bf16 in registers
// (uir_mfma_split, f32-accurate); 0 = on the f32 MFMA (the round-1..3 arithmetic, kept for A/B measurements and the parity tests)
int uir_fused_split() {
  static const int mode = [] {
    const char* v = getenv("UIR_FUSED_SPLIT");
    return (v && v[0] == '0' && !v[1]) ? 0 : 1;
  }();
  return mode;
}

int uir_tiled_conv3x3(const UirTiledArgs& t, hipStream_t st) {
  UIR_REQUIRE(t.A && t.B && t.D && t.MM > 0 && t.NN > 0 && t.KK > 0 && (t.KK & 3) == 0 && t.sa_k == 1 && (t.sa_i & 3) == 0);
  UIR_REQUIRE(t.H > 0 && t.Wd > 0 && t.MM % (t.H * t.Wd) == 0 && uir_aligned16(t.A) && t.act == UIR_ACT_NONE && !t.pre_out);
  const dim3 grid(((t.MM + TM - 1) / TM) * ((t.NN + TN - 1) / TN));
  hipLaunchKernelGGL(uir_tiled_conv_kernel, grid, dim3(256), (size_t)4 * STAGE * sizeof(float), st, t);
  return uir_launch_status();
}

int uir_tiled_gemm(const UirTiledArgs& t_in, hipStream_t st) {
  UirTiledArgs t = t_in;
  // row-major float4 epilogue (uir_rowmajor_epilogue) wherever the output rows, the residual, the pre-activation copy and the bias are
  // 16-byte addressable; the transposed weight-gradient layouts (sd_j != 1) keep the column-per-lane stores
  t.vec_epi = t.sd_j == 1 && (t.sd_i & 3) == 0 && (t.NN & 3) == 0 && (t.d_split & 3) == 0 && uir_aligned16(t.D) &&
              (!t.R || ((t.ldr & 3) == 0 && uir_aligned16(t.R))) && (!t.pre_out || ((t.ldpre & 3) == 0 && uir_aligned16(t.pre_out))) &&
              (!t.bias || uir_aligned16(t.bias));
  UIR_REQUIRE(t.A && t.B && t.D && t.MM > 0 && t.NN > 0 && t.KK > 0 && t.k_per_split > 0 && (t.k_per_split & 3) == 0 && (t.KK & 3) == 0);
  const bool akc = t.sa_k == 1, bkc = t.sb_k == 1;
  UIR_REQUIRE(akc || t.sa_i == 1);
  UIR_REQUIRE(bkc || t.sb_j == 1);
  UIR_REQUIRE(uir_aligned16(t.A) && uir_aligned16(t.B));
  if (akc) UIR_REQUIRE((t.sa_i & 3) == 0); else UIR_REQUIRE((t.sa_k & 3) == 0 && (t.MM & 3) == 0 && t.MM >= 4);
  if (bkc) UIR_REQUIRE((t.sb_j & 3) == 0); else UIR_REQUIRE((t.sb_k & 3) == 0 && (t.NN & 3) == 0 && t.NN >= 4);
  UIR_REQUIRE(t.KK >= 4);
  const int nsplit = (t.KK + t.k_per_split - 1) / t.k_per_split;
  const dim3 grid(((t.MM + TM - 1) / TM) * ((t.NN + TN - 1) / TN), nsplit);
  const size_t lds = (size_t)4 * STAGE * sizeof(float);
  static_assert(STAGE >= TK * IP, "stage size");
  // Split-bf16 operands on the bf16 matrix pipe (uir_tiled_split_kernel: three bf16 planes per f32 operand, six products, f32
  // accumulation - as accurate against fp64 as the f32 MFMA, tests/test_tiled.py).  Since its loads overlap its MFMAs (round 3) it
  // beats the f32 kernel wherever the tile grid fills the chip and the contraction is long enough to amortise the split:
  //   UIR_TILED_SPLIT unset / 2 (default): AUTO - full 128 x 128 tiles (M, N >= 128), >= 256 workgroups, >= 64 k per split
  //                   (16 k x 512 -> 2048: 295 -> 212 us = 162 TFLOP/s, above the f32 MFMA peak; 65 k x 256 -> 1024: 328 -> 229 us);
  //                   narrow outputs, under-filled grids (the SpectralTransformer's own four deep-level calls) and short
  //                   contractions stay on the f32 kernel and its 128 x 64 / 64 x 128 variants
  //   0: f32 MFMA everywhere;  1: every tiled call;  3: only the calls the dispatchers mark (t.allow_split)
  // Waves issuing v_mfma_f32_32x32x16_bf16 corrupt packed-f32 instructions with a half-swapped src1 in ANY wave sharing their SIMD
  // (gfx950 erratum): build.py's ISA lint keeps that instruction form out of every kernel of this library.
  static const int mode = getenv("UIR_TILED_SPLIT") ? atoi(getenv("UIR_TILED_SPLIT")) : 2;
  UIR_REQUIRE(mode >= 0 && mode <= 3);
  const int kc_split = t.k_per_split < t.KK ? t.k_per_split : t.KK;
  const bool auto_split = mode == 2 && t.MM >= TM && t.NN >= TN && kc_split >= 64 && (long)grid.x * nsplit >= 256;
  // UIR_MIXED=bf16 | fp16 (opt-in, knobs.py): EVERY contraction that reaches this kernel family runs on uir_tiled_lp_kernel - operands
  // rounded to the named 16-bit type, f32 accumulation and epilogue.  Results are no longer f32-accurate (that is the point of the mode)
  // and no f32 number of this repository is measured with it.
  const int mixed = uir_mixed_mode();
  UIR_REQUIRE(mixed >= 0 && !(mixed && (mode == 1 || mode == 3)));
  if (mixed) {
    const size_t ldsm = std::max((size_t)4 * SPLANE * sizeof(unsigned short), (size_t)4 * 32 * EPW * sizeof(float));
#define UIR_LP(AK, BK) do { if (mixed == 2) hipLaunchKernelGGL((uir_tiled_lp_kernel<AK, BK, true>), grid, dim3(256), ldsm, st, t); \
                            else hipLaunchKernelGGL((uir_tiled_lp_kernel<AK, BK, false>), grid, dim3(256), ldsm, st, t); } while (0)
    if (akc && bkc) UIR_LP(true, true);
    else if (akc) UIR_LP(true, false);
    else if (bkc) UIR_LP(false, true);
    else UIR_LP(false, false);
#undef UIR_LP
    return uir_launch_status();
  }
  if ((!t.out_scale || t.vec_epi) && (mode == 1 || (mode == 3 && t.allow_split) || auto_split)) {
    const size_t lds2 = (size_t)2 * SSTAGE * sizeof(unsigned short);
#define UIR_TS(AK, BK) hipLaunchKernelGGL((uir_tiled_split_kernel<AK, BK>), grid, dim3(256), lds2, st, t)
    if (akc && bkc) UIR_TS(true, true);
    else if (akc) UIR_TS(true, false);
    else if (bkc) UIR_TS(false, true);
    else UIR_TS(false, false);
#undef UIR_TS
    return uir_launch_status();
  }
  // narrow second operand (<= 64 columns, or 65 .. 127 where half of a 128-wide tile would be empty... only <= 64 here): 128 x 64 tiles
  if (t.NN <= 64) {
    const dim3 grid64(((t.MM + TM - 1) / TM) * ((t.NN + 63) / 64), nsplit);
#define UIR_TL64(AK, BK) hipLaunchKernelGGL((uir_tiled_kernel<AK, BK, 64>), grid64, dim3(256), lds, st, t)
    if (akc && bkc) UIR_TL64(true, true);
    else if (akc) UIR_TL64(true, false);
    else if (bkc) UIR_TL64(false, true);
    else UIR_TL64(false, false);
#undef UIR_TL64
    return uir_launch_status();
  }
  // few row tiles (the deep levels of the SpectralTransformer: 16 384 rows x 128 columns = 128 tiles on 256 CUs): 64-row tiles
  if ((long)grid.x * nsplit < 256 && t.MM > 64) {
    const dim3 gridh(((t.MM + 63) / 64) * ((t.NN + TN - 1) / TN), nsplit);
#define UIR_TLH(AK, BK) hipLaunchKernelGGL((uir_tiled_kernel<AK, BK, 128, 64>), gridh, dim3(256), lds, st, t)
    if (akc && bkc) UIR_TLH(true, true);
    else if (akc) UIR_TLH(true, false);
    else if (bkc) UIR_TLH(false, true);
    else UIR_TLH(false, false);
#undef UIR_TLH
    return uir_launch_status();
  }
#define UIR_TL(AK, BK) hipLaunchKernelGGL((uir_tiled_kernel<AK, BK, 128>), grid, dim3(256), lds, st, t)
  if (akc && bkc) UIR_TL(true, true);
  else if (akc) UIR_TL(true, false);
  else if (bkc) UIR_TL(false, true);
  else UIR_TL(false, false);
#undef UIR_TL
  return uir_launch_status();
}
