// uir_gram_split: the pixel contractions of the deep levels' weight gradients (dW_in = dT1^T LN(x), dW_qkv = dQKV^T LN(x), dW_out = dy^T gate(T2) at
// C = 64 / 128: SpectralTransformer.py:81,122,125 under autograd) and the other plain 1x1 weight gradients of the step on split-bf16 MFMA.
//
//     out[z][n][k] = sum_m A[z][m][n] * pro(B[z][m][k])          uir_gram's contract (taps = 1, no column extras), same partial-slab rows
//
// On csrc/gram.hip these run on the f32 MFMA with one dword load per operand element and MFMA step (22-95 TFLOP/s, 1.1-1.8 TB/s): at levels
// 2 and 3 they cost the step their own kernel time one for one (timing-only skip experiment, profiles/r05).  Here both operands are staged
// through LDS as three bf16 planes (x = h + m + l exactly; prologue - LayerNorm from the saved statistics, or the GDFN gate - applied
// when the element is staged, ONCE per workgroup) in their own pixel-major [pixel][channel] layout with 16-byte loads, and the waves read
// them TRANSPOSED (ds_read_b64_tr_b16: a lane gets four consecutive pixels of one channel), which is the k-contiguous fragment the
// 32x32x16 MFMA wants for a contraction over pixels.  Six bf16 MFMAs per 16-pixel step reproduce the f32 product to 3 * 2^-24.
// A workgroup owns a 128 x QB block (QB = 64 or 128: "P" is the operand with more columns) for its chunk of rows.
#include "uir_common.h"
#include "../../include/uir_kernels.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void split3(float x, uint32_t& h, uint32_t& m, uint32_t& l) {
  h = __float_as_uint(x) & 0xffff0000u;
  const float r1 = x - __uint_as_float(h);
  m = __float_as_uint(r1) & 0xffff0000u;
  l = __float_as_uint(r1 - __uint_as_float(m));
}
template <int PL>
__device__ __forceinline__ void store_quad(unsigned short* d, const float (&x)[4]) {
  uint32_t h[4], m[4], l[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) split3(x[e], h[e], m[e], l[e]);
  *reinterpret_cast<uint2*>(d) = make_uint2(uir_pack_hi16(h[0], h[1]), uir_pack_hi16(h[2], h[3]));
  *reinterpret_cast<uint2*>(d + PL) = make_uint2(uir_pack_hi16(m[0], m[1]), uir_pack_hi16(m[2], m[3]));
  *reinterpret_cast<uint2*>(d + 2 * PL) = make_uint2(uir_pack_hi16(l[0], l[1]), uir_pack_hi16(l[2], l[3]));
}

// LDS image: pixel rows of ROW bf16 (64 -> 128 bytes, 128 -> 256 bytes).  A half-wave's transposed read takes 4 consecutive pixel rows x
// 64 bytes (one 32-channel tile): the 64-byte windows of consecutive rows must fall on different quarters of the 256-byte bank line.
template <int ROW>
__device__ __forceinline__ int wsz(int row, int ch) {           // ch = channel (multiple of 4) -> bf16 offset
  int chunk = ch >> 3;                                           // 16-byte chunk of the row
  if (ROW == 64) chunk ^= ((row >> 1) & 1) << 2;                 // rows r, r + 2 share a window position: swap the halves
  else chunk ^= (row & 3) << 2;                                  // 256-byte rows: every row starts a bank line: rotate the four windows
  return row * ROW + (chunk << 3) + (ch & 7);
}
__device__ __forceinline__ s16x4 tr_read(const unsigned short* p) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p));
}
__device__ __forceinline__ bf16x8 pack8(s16x4 lo, s16x4 hi) {
  struct { s16x4 a, b; } t = {lo, hi};
  return __builtin_bit_cast(bf16x8, t);
}

struct GbArgs {
  const float* P; const float* Q;      // operand with the 128-column block / the QB-column block
  float* slab;
  const float* stats; const float* ln_g; const float* ln_b;     // PRO 1 (LayerNorm): per-row (mean, rstd), per-column gamma / beta
  long p_bs, q_bs, o_sp, o_sq, o_bs, slab_stride;
  int ldp, ldq, np, nq;                // row strides, padded column counts
  int p_real, p_pad, q_real, q_pad;    // column -> real output index (groups of pad columns, the first real ones real; pad <= 0: identity)
  int gate_off;                        // PRO 2 (gate): value = gelu(x[c]) * x[gate_off + c]
  int M, rows_per_chunk;
};

// one operand's stage: SR rows x COLS columns from row0 (rows >= row_end and columns >= ncols give zeros), prologue PRO, split, LDS.
// NT threads, thread t owns the float4 at column 4 (t % QPR) of rows t / QPR + RPI i: its column, column mask, LayerNorm gamma / beta and the
// swizzle term of its LDS address are the same for every i and every stage (RPI is a multiple of 4), so a stage costs per float4 the
// prologue + split + three 8-byte LDS stores and nothing else (first written with per-element masks and per-float4 address arithmetic the
// staging was 1 287 VALU instructions per stage and wave against 48 MFMAs: the kernel ran slower than the f32 one it replaces).
template <int COLS, int SR, int PRO, int NT>
struct Stager {
  static constexpr int QPR = COLS / 4;                           // float4 per row
  static constexpr int RPI = NT / QPR;                           // rows between a thread's consecutive float4
  static constexpr int NQ = SR / RPI;
  static constexpr int PL = SR * COLS;                           // bf16 per plane
  static_assert(NT % QPR == 0 && SR % RPI == 0 && RPI % 4 == 0, "thread -> (row, column) map");
  uint4 v[NQ], g[PRO == 2 ? NQ : 1];
  float2 st[PRO == 1 ? NQ : 1];
  const float* src;              // this thread's column of row 0 (clamped column)
  const float* stats;            // + 2 * its first row
  float4 ga, be;                 // PRO 1
  uint32_t cm[4];                // column masks: all ones for a real column
  int r0, lds0, ld, goff;
  __device__ __forceinline__ void init(const float* base, int ld_, int col0, int ncols, const float* stats_, int gate_off, const float* ln_g,
                                       const float* ln_b, int tid) {
    r0 = tid / QPR;
    const int cl = 4 * (tid - r0 * QPR), c = col0 + cl, cc = min(c, (ncols - 1) & ~3);   // clamped: always readable (ld >= roundup4(ncols))
    ld = ld_; goff = gate_off;
    src = base + cc;
    stats = stats_;
    lds0 = wsz<COLS>(r0, cl);
#pragma unroll
    for (int k = 0; k < 4; ++k) cm[k] = c + k < ncols ? 0xffffffffu : 0u;
    if constexpr (PRO == 1) { ga = *reinterpret_cast<const float4*>(ln_g + cc); be = *reinterpret_cast<const float4*>(ln_b + cc); }
  }
  __device__ __forceinline__ void load(int row0, int row_end) {
    if (row0 + SR <= row_end) {
      const float* p = src + (long)(row0 + r0) * ld;
#pragma unroll
      for (int i = 0; i < NQ; ++i) {
        v[i] = *reinterpret_cast<const uint4*>(p + (long)i * RPI * ld);
        if constexpr (PRO == 2) g[i] = *reinterpret_cast<const uint4*>(p + (long)i * RPI * ld + goff);
        if constexpr (PRO == 1) st[i] = *reinterpret_cast<const float2*>(stats + 2L * (row0 + r0 + i * RPI));
      }
    } else {                     // the chunk's last, partial stage: clamped rows (masked when stored)
#pragma unroll
      for (int i = 0; i < NQ; ++i) {
        const int rr = min(row0 + r0 + i * RPI, row_end - 1);
        const float* p = src + (long)rr * ld;
        v[i] = *reinterpret_cast<const uint4*>(p);
        if constexpr (PRO == 2) g[i] = *reinterpret_cast<const uint4*>(p + goff);
        if constexpr (PRO == 1) st[i] = *reinterpret_cast<const float2*>(stats + 2L * rr);
      }
    }
  }
  __device__ __forceinline__ void store(unsigned short* s, int row0, int row_end) const {
    const bool whole = row0 + SR <= row_end;
#pragma unroll
    for (int i = 0; i < NQ; ++i) {
      float x[4] = {__uint_as_float(v[i].x), __uint_as_float(v[i].y), __uint_as_float(v[i].z), __uint_as_float(v[i].w)};
      if constexpr (PRO == 1) {
        const float mu = st[i].x, rs = st[i].y;
        x[0] = fmaf((x[0] - mu) * rs, ga.x, be.x); x[1] = fmaf((x[1] - mu) * rs, ga.y, be.y);
        x[2] = fmaf((x[2] - mu) * rs, ga.z, be.z); x[3] = fmaf((x[3] - mu) * rs, ga.w, be.w);
      }
      if constexpr (PRO == 2) {
        x[0] = uir_gelu(x[0]) * __uint_as_float(g[i].x); x[1] = uir_gelu(x[1]) * __uint_as_float(g[i].y);
        x[2] = uir_gelu(x[2]) * __uint_as_float(g[i].z); x[3] = uir_gelu(x[3]) * __uint_as_float(g[i].w);
      }
      const uint32_t rm = (whole || row0 + r0 + i * RPI < row_end) ? 0xffffffffu : 0u;
#pragma unroll
      for (int k = 0; k < 4; ++k) x[k] = __uint_as_float(__float_as_uint(x[k]) & cm[k] & rm);
      store_quad<PL>(s + lds0 + i * RPI * COLS, x);
    }
  }
};

// PROP / PROQ: prologue of the P / Q operand (0 none, 1 LayerNorm, 2 gate).  Waves (2 P pairs) x (2 Q halves): each 64 x 32 (QB = 64) or 64 x 64 (QB = 128).
// 32-row stages: a stage costs one trip to memory (the next stage's loads are requested right behind the barrier, a stage's 48-96 MFMAs are far
// shorter than that trip), so what the kernel moves is set by how many stages a CU has in flight - four workgroups of 36-48 KB at <= 128
// registers (64-row stages on two workgroups per CU ran 3.5-8 us per stage: slower than the f32 kernel it replaces).
template <int QB, int PROP, int PROQ>
__global__ __launch_bounds__(256, QB == 64 ? 4 : 3) void uir_gramb_kernel(const GbArgs a, int qblocks) {
  constexpr int SR = 32;
  constexpr int QT = QB / 64;                                    // Q tiles per wave
  constexpr int PLP = SR * 128, PLQ = SR * QB;
  extern __shared__ float smem[];
  unsigned short* Ps = reinterpret_cast<unsigned short*>(smem);
  unsigned short* Qs = Ps + 3 * PLP;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, h2 = lane >> 5;
  const int pw = wave >> 1, qw = wave & 1;
  const int z = blockIdx.z;
  const int pb = (int)blockIdx.y / qblocks, qb = (int)blockIdx.y % qblocks;
  const int p0 = 128 * pb, q0 = QB * qb;
  const float* P = a.P + (long)z * a.p_bs;
  const float* Q = a.Q + (long)z * a.q_bs;
  const float* stats = (PROP == 1 || PROQ == 1) ? a.stats + (long)z * a.M * 2 : nullptr;
  const int row_begin = (int)blockIdx.x * a.rows_per_chunk;
  const int row_end = min(a.M, row_begin + a.rows_per_chunk);
  const int G = lane >> 4, q = (lane >> 2) & 3, c4 = 4 * (lane & 3);
  const int pcol = 64 * pw + 16 * (G & 1) + c4, qcol = 32 * QT * qw + 16 * (G & 1) + c4;

  f32x16 acc[2][QT];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < QT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  Stager<128, SR, PROP, 256> sp;
  Stager<QB, SR, PROQ, 256> sq;
  sp.init(P, a.ldp, p0, a.np, stats, a.gate_off, a.ln_g, a.ln_b, tid);
  sq.init(Q, a.ldq, q0, a.nq, stats, a.gate_off, a.ln_g, a.ln_b, tid);
  if (row_begin < row_end) {
    sp.load(row_begin, row_end);
    sq.load(row_begin, row_end);
  }
  for (int row0 = row_begin; row0 < row_end; row0 += SR) {
    sp.store(Ps, row0, row_end);
    sq.store(Qs, row0, row_end);
    __syncthreads();
    const int nxt = row0 + SR < row_end ? row0 + SR : row0;          // next stage in flight during the MFMAs (clamped, unconditional)
    sp.load(nxt, row_end);
    sq.load(nxt, row_end);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int s = 0; s < SR / 16; ++s) {
      // k slot (h2, e) of the 16-deep step = stage row 16 s + 8 h2 + e; this lane addresses block rows q and q + 4 of its 16-lane group
      const int r0 = 16 * s + 8 * h2 + q;
      bf16x8 fp[2][3], fq[QT][3];
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
          fp[i][pl] = pack8(tr_read(Ps + pl * PLP + wsz<128>(r0, pcol + 32 * i)), tr_read(Ps + pl * PLP + wsz<128>(r0 + 4, pcol + 32 * i)));
#pragma unroll
        for (int j = 0; j < QT; ++j)
          fq[j][pl] = pack8(tr_read(Qs + pl * PLQ + wsz<QB>(r0, qcol + 32 * j)), tr_read(Qs + pl * PLQ + wsz<QB>(r0 + 4, qcol + 32 * j)));
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < QT; ++j) {
          f32x16 c = acc[i][j];
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fp[i][1], fq[j][1], c, 0, 0, 0);     // small terms first
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fp[i][2], fq[j][0], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fp[i][0], fq[j][2], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fp[i][1], fq[j][0], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fp[i][0], fq[j][1], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fp[i][0], fq[j][0], c, 0, 0, 0);
          acc[i][j] = c;
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();                                    // every wave has read this stage
  }

  // D[i][j]: i = P column (accumulator row), j = Q column (lane)
  float* out = a.slab + (long)blockIdx.x * a.slab_stride + (long)z * a.o_bs;
#pragma unroll
  for (int j = 0; j < QT; ++j) {
    const int qc = q0 + 32 * (QT * qw + j) + li;
    const int qr = qc < a.nq ? uir_colmap(qc, a.q_real, a.q_pad) : -1;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int pc = p0 + 64 * pw + 32 * i + uir_mfma_row(r, h2);
        const int pr = pc < a.np ? uir_colmap(pc, a.p_real, a.p_pad) : -1;
        if (pr >= 0 && qr >= 0) out[(long)pr * a.o_sp + (long)qr * a.o_sq] = acc[i][j][r];
      }
  }
}

template <int QB, int PROP, int PROQ>
int gb_launch(const GbArgs& g, int chunks, int pblocks, int qblocks, int batches, hipStream_t st) {
  constexpr int SR = 32;
  const size_t lds = (size_t)3 * SR * (128 + QB) * sizeof(unsigned short);
  hipLaunchKernelGGL((uir_gramb_kernel<QB, PROP, PROQ>), dim3(chunks, pblocks * qblocks, batches), dim3(256), lds, st, g, qblocks);
  return uir_launch_status();
}

}  // namespace

// -1: not this kernel's case (uir_gram continues on its f32 MFMA paths)
int uir_gram_split(const UirGramArgs& a, void* stream) {
  if (!uir_fused_split() || uir_mixed_mode() != 0) return -1;
  if (a.taps != 1 || a.sq_a_off >= 0 || a.sq_b_off >= 0 || a.cs_a_off >= 0) return -1;
  // long contractions of at least one 32-column tile on each side; the wide x wide weight gradients of the other networks keep
  // csrc/tiled.hip's 128 x 128 tiles (this kernel re-reads an operand once per block of the other)
  if (a.M < 1024 || a.N < 32 || a.K < 32 || (a.N >= 256 && a.K >= 256)) return -1;
  // rows are read in float4 units up to roundup4(columns): a column count that is no multiple of 4 (dW_out: K = h) needs the room in its row
  const int n4 = (a.N + 3) & ~3, k4 = (a.K + 3) & ~3;
  if (((a.lda | a.ldb) & 3) || a.lda < n4 || a.ldb < k4 || !uir_aligned16(a.A) || !uir_aligned16(a.B) || (a.a_bs & 3) || (a.b_bs & 3)) return -1;
  if (a.pro_b == UIR_GRAM_PRO_GATE && ((a.gate_off & 3) || a.gate_off < k4 || a.ldb < a.gate_off + k4)) return -1;
  if (a.pro_b == UIR_GRAM_PRO_LN && ((a.K & 3) || !uir_aligned16(a.ln_g) || !uir_aligned16(a.ln_b) || !a.stats)) return -1;
  // a chunk = whole 32-row stages (engine.Recorder._chunks shapes them so for this kernel; anything else keeps the f32 kernels' chunking)
  if ((long)a.rows_per_chunk * a.nchunks < a.M || (a.rows_per_chunk & 31)) return -1;
  const bool swap = a.K > a.N;                 // P = the operand with more columns
  // prologue on P only as the gate (dW_out: K = h > N = C), on Q only as LayerNorm (dW_in / dW_qkv: K = C < N)
  if (a.pro_b == UIR_GRAM_PRO_GATE && !swap) return -1;
  if (a.pro_b == UIR_GRAM_PRO_LN && swap) return -1;
  GbArgs g{};
  g.slab = a.slab; g.slab_stride = a.slab_stride; g.o_bs = a.o_bs;
  g.stats = a.stats; g.ln_g = a.ln_g; g.ln_b = a.ln_b; g.gate_off = a.gate_off;
  g.M = a.M; g.rows_per_chunk = a.rows_per_chunk;
  if (!swap) {
    g.P = a.A; g.ldp = a.lda; g.np = a.N; g.p_bs = a.a_bs; g.p_real = a.n_grp_real; g.p_pad = a.n_grp_pad; g.o_sp = a.o_sn;
    g.Q = a.B; g.ldq = a.ldb; g.nq = a.K; g.q_bs = a.b_bs; g.q_real = a.k_grp_real; g.q_pad = a.k_grp_pad; g.o_sq = a.o_sk;
  } else {
    g.P = a.B; g.ldp = a.ldb; g.np = a.K; g.p_bs = a.b_bs; g.p_real = a.k_grp_real; g.p_pad = a.k_grp_pad; g.o_sp = a.o_sk;
    g.Q = a.A; g.ldq = a.lda; g.nq = a.N; g.q_bs = a.a_bs; g.q_real = a.n_grp_real; g.q_pad = a.n_grp_pad; g.o_sq = a.o_sn;
  }
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int pblocks = (g.np + 127) / 128;
  const bool q128 = g.nq > 64;
  const int qblocks = q128 ? (g.nq + 127) / 128 : 1;
#define UIR_GB(QBV) do { \
    if (a.pro_b == UIR_GRAM_PRO_GATE) return gb_launch<QBV, 2, 0>(g, a.nchunks, pblocks, qblocks, a.batches, st); \
    if (a.pro_b == UIR_GRAM_PRO_LN) return gb_launch<QBV, 0, 1>(g, a.nchunks, pblocks, qblocks, a.batches, st); \
    return gb_launch<QBV, 0, 0>(g, a.nchunks, pblocks, qblocks, a.batches, st); } while (0)
  if (q128) UIR_GB(128);
  UIR_GB(64);
#undef UIR_GB
}
