// Window attention core of WindowAttention_Sparse (reference src/model/block.py:284-370), forward and
// backward, for the 8 x 8 windows every DecoderBlock / MDASSA of NewBigFRFNModel uses (model.py:517-...: win_size 8,
// shift_size 0, mask None) and, with args.shift = 4, the shifted windows + region mask of AST's TransformerBlock
// (src/Models/AST.py:144-219,568-616: same attention arithmetic, scores + mask before softmax and relu):
//   attn = (q * scale) k^T + relative_position_bias            [heads, 64, 64] per window          (:340-352)
//   attn = softmax(attn) * w1 + relu(attn)^2 * w2,  (w1, w2) = softmax(self.w)                      (:361-367)
//   out  = attn v, heads concatenated along the channels                                             (:370)
// The Linear projections around it (to_q / to_kv / proj, :322,327,371) are uir_gemm calls on the token rows: a window is just a
// set of 64 rows of the HWC token image, gathered here by address - window_partition / window_reverse (:16-41) never move data.
//
// Round 3: every product of the core is a v_mfma_f32_32x32x2_f32 chain and NOTHING but three small tables goes through LDS.
// One WAVE owns one (window, head); four waves per workgroup, no workgroup barriers.  The 64 x 64 score matrix is held TRANSPOSED in
// accumulator registers - S^T = K Q^T, key on the accumulator row, query on the lane - so that
//   * a query's softmax runs over the lane's own registers plus one v_permlane32_swap (the two lane halves hold 16 keys each);
//   * the accumulator registers ARE the A operand of the next product (lane = row index, lane half = contraction index): O = P V,
//     dQ = dS K consume P / dS straight from the registers they were computed in; the matching B operand is V / K read "channel on
//     the lane" for the two key tokens the register stands for.
// dK = dS^T Q and dV = P^T dO contract over the QUERIES, which sit on the lanes of that layout: the backward therefore recomputes the
// score / dP tiles in the other orientation (S = Q K^T, query on the row, key on the lane; the per-query softmax statistics travel
// through 768 bytes of LDS) instead of transposing two 64 x 64 matrices through 33 KB of LDS, which capped the first version at three
// waves per CU (0.75 per SIMD: 1.25 ms per launch at 4 x 512 x 512, C = 32, against 0.2 ms of HBM time).  448 MFMAs per (window, head)
// at head dim 32; operands come from global memory / L1 in both layouts (token-major float4 runs for the score products, one dword
// per lane for the channel-major B operands).
// relative_position_index (:302-311) is evaluated from the token coordinates: in either orientation it is (a per-lane base) +- (a
// compile-time constant of the register), i.e. an immediate LDS offset.  Backward writes dQ / dK / dV rows and per-(window, head)
// partials of the bias-table and mixing-weight gradients (summed in fixed order by uir_reduce_slab(_batch); the table partial is
// accumulated with plain LDS read / add / write sequences - the addresses are distinct inside each instruction and a wave's LDS
// instructions execute in order - on four wave-private tables, since round 5: order fixed, bitwise run-to-run; as ds_add_f32 they held
// the LDS pipe 78 % of the launch).
#include "uir_common.h"
#include "../../include/uir_kernels.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int WS = 8, NT = WS * WS, RP = 2 * WS - 1, NTAB = RP * RP;   // window side, tokens per window, relative positions
constexpr int TABP = 228;                                              // padded table length

__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0); }
__device__ __forceinline__ float swap32_sum(float x) {        // lanes l and l ^ 32
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float swap32_max(float x) {
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
// phase boundary for the scheduler: straight-line at head dim <= 32, hipcc otherwise hoists the operand loads of later products above
// the softmax (256 VGPRs + 95 spilled); with the boundaries the backward needs < 200
__device__ __forceinline__ void phase() { asm volatile("" ::: "memory"); }
__device__ __forceinline__ void lds_wait() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }   // wave-local LDS hand-over

template <int HD>
struct WA {
  static constexpr int CK = HD < 32 ? HD : 32;      // channels per contraction chunk of the score products
  static constexpr int HK = CK / 2;                 // ... per lane half (a contiguous run of the token's row)
  static constexpr int NCK = HD / CK;
  static constexpr int NCT = (HD + 31) / 32;        // 32-channel column tiles of the value-side products
};

// Geometry of one window for one wave.  Token (ty, tx) of the window is image pixel ((wy 8 + ty + shift) mod H, (wx 8 + tx + shift)
// mod W): the cyclic roll of the shifted windows is an address.
struct WinGeo {
  int l32, half;
  int pix_tm[2];     // token-major layout: pixel of token tile * 32 + l32
  int rowp[8];       // wave-uniform: pixel index of image column 0 in the row of window row ty
  int xcol[4];       // image column of window column half * 4 + e
  int region_q[2];   // shift region of the lane's token in tile 0 / 1 (0 without shift)
};

__device__ __forceinline__ int region_of(const UirWinAttnArgs& a, int wy, int wx, int ty, int tx) {
  // AST.py:571-582: three row bands x three column bands of the ROLLED image, cut at n - ws and n - shift
  const int y = wy * WS + ty, x = wx * WS + tx;
  const int ry = y < a.H - WS ? 0 : (y < a.H - a.shift ? 1 : 2);
  const int rx = x < a.W - WS ? 0 : (x < a.W - a.shift ? 1 : 2);
  return ry * 3 + rx;
}

__device__ __forceinline__ void make_geo(const UirWinAttnArgs& a, int win, int lane, WinGeo& g, int* regs) {
  const int wpr = a.W / WS, wpi = (a.H / WS) * wpr;
  const int b = win / wpi, r = win - b * wpi;
  const int wy = r / wpr, wx = r - wy * wpr;
  g.l32 = lane & 31;
  g.half = lane >> 5;
#pragma unroll
  for (int ty = 0; ty < 8; ++ty) {
    int y = wy * WS + ty + a.shift;
    y -= y >= a.H ? a.H : 0;
    g.rowp[ty] = (b * a.H + y) * a.W;
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    int x = wx * WS + g.half * 4 + e + a.shift;
    x -= x >= a.W ? a.W : 0;
    g.xcol[e] = x;
  }
  int xl = wx * WS + (lane & 7) + a.shift;
  xl -= xl >= a.W ? a.W : 0;
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int ty = t * 4 + (g.l32 >> 3);
    int y = wy * WS + ty + a.shift;
    y -= y >= a.H ? a.H : 0;
    g.pix_tm[t] = (b * a.H + y) * a.W + xl;
    g.region_q[t] = a.shift > 0 ? region_of(a, wy, wx, ty, lane & 7) : 0;
  }
  regs[lane] = a.shift > 0 ? region_of(a, wy, wx, lane >> 3, lane & 7) : 0;     // region of window token `lane`
}

template <int HK>
__device__ __forceinline__ void ld_run(const float* __restrict__ base, unsigned off, float (&v)[HK]) {
#pragma unroll
  for (int c = 0; c < HK; c += 4) {
    const float4 t = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(base) + (off + c) * 4u);   // s[base] + v_off
    v[c] = t.x; v[c + 1] = t.y; v[c + 2] = t.z; v[c + 3] = t.w;
  }
}

// d0 = A[tile 0] . B[tile tb]^T, d1 = A[tile 1] . B[tile tb]^T over head h's HD channels (rows of d = tokens of A, lanes = tokens of B)
template <int HD>
__device__ __forceinline__ void score_pair(const float* __restrict__ A, int lda, const float* __restrict__ Bm, int ldb, int tb,
                                           const WinGeo& g, int h, f32x16& d0, f32x16& d1) {
  using T = WA<HD>;
#pragma unroll
  for (int r = 0; r < 16; ++r) { d0[r] = 0.f; d1[r] = 0.f; }
  // the trip count is made opaque: a one-trip loop (head dim <= 32) is flattened into straight-line code, which hipcc then schedules
  // with every operand load at the top (256 VGPRs + spills against 189 for the identical code at head dim 64)
  int nck = T::NCK;
  asm volatile("" : "+s"(nck));
#pragma unroll 1
  for (int ck = 0; ck < nck; ++ck) {
    const int col = h * HD + ck * T::CK + g.half * T::HK;
    float a0[T::HK], a1[T::HK], b[T::HK];
    ld_run<T::HK>(Bm, (unsigned)(tb ? g.pix_tm[1] : g.pix_tm[0]) * ldb + col, b);   // (a runtime index would put the array in scratch)
    ld_run<T::HK>(A, (unsigned)g.pix_tm[0] * lda + col, a0);
    ld_run<T::HK>(A, (unsigned)g.pix_tm[1] * lda + col, a1);
#pragma unroll
    for (int s = 0; s < T::HK; ++s) {
      d0 = mfma32(a0[s], b[s], d0);
      d1 = mfma32(a1[s], b[s], d1);
    }
  }
}

// out[m][n] = sum over the 64 tokens t of p[m][t] X[t][ct * 32 + n]: p0 / p1 are accumulator tiles whose ROW index is the token
// (tile 0 / 1), i.e. register r of a lane stands for token tile * 32 + (r / 4) * 8 + half * 4 + r % 4, and the lane's column of the tile
// becomes the output row.  X is read channel-on-the-lane for exactly those tokens.
// PIN (the forward kernel): all 32 operand loads are issued before the first MFMA.  Left to itself hipcc sinks each load next to the
// MFMA that consumes it (fewer live registers: 97) - 32 serialised `global_load_dword; s_waitcnt vmcnt(0); v_mfma` round trips per
// column tile; the backward kernel's uses are batched by the compiler already and stay as they are.
template <int HD, bool PIN = false>
__device__ __forceinline__ void acc_times(const f32x16& p0, const f32x16& p1, const float* __restrict__ X, int ldx, const WinGeo& g,
                                          int h, int ct, f32x16& out) {
  // head dim < 32: lanes past the head's channels re-read its last channel (their output columns are not stored).  The column keeps
  // its dependence on ct there too: loop-invariant, the loads of both products of a tile loop were hoisted in front of it (+64 VGPRs)
  const int cl = ct * 32 + g.l32;
  const int col = h * HD + (HD < 32 ? (cl < HD ? cl : HD - 1) : cl);
#pragma unroll
  for (int r = 0; r < 16; ++r) out[r] = 0.f;
  // uniform row base (scalar registers) + one of four per-lane byte offsets: `global_load_dword v, v_off, s[base]`, no 64-bit VALU
  // address per load (and nothing for the compiler to hoist across the tile loops: 32 tokens x 6 tensors of addresses otherwise)
  unsigned vo[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) vo[e] = ((unsigned)g.xcol[e] * ldx + col) * 4u;
  float b[32];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const char* rb = reinterpret_cast<const char*>(X) + (size_t)((unsigned)g.rowp[t * 4 + r / 4] * ldx) * 4u;
      b[t * 16 + r] = *reinterpret_cast<const float*>(rb + vo[r % 4]);
    }
  if (PIN) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int r = 0; r < 16; ++r) out = mfma32(p0[r], b[r], out);
#pragma unroll
  for (int r = 0; r < 16; ++r) out = mfma32(p1[r], b[16 + r], out);
}

// rows of `out` = tokens of tile `tile` (register r <-> token (r / 4) * 8 + half * 4 + r % 4), lanes = channels ct * 32 + l32
template <int HD>
__device__ __forceinline__ void store_rows(float* __restrict__ D, int ldd, const WinGeo& g, int h, int ct, int tile, const f32x16& out,
                                           float mul) {
  if (HD < 32 && g.l32 >= HD) return;
  const int col = h * HD + ct * 32 + g.l32;
  unsigned vo[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) vo[e] = ((unsigned)g.xcol[e] * ldd + col) * 4u;
  int rp[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) rp[j] = tile ? g.rowp[4 + j] : g.rowp[j];
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    char* rb = reinterpret_cast<char*>(D) + (size_t)((unsigned)rp[r / 4] * ldd) * 4u;
    *reinterpret_cast<float*>(rb + vo[r % 4]) = out[r] * mul;
  }
}

__device__ __forceinline__ void mix_weights(const float* w, float& w1, float& w2) {
  const float m = fmaxf(w[0], w[1]);
  const float e0 = __expf(w[0] - m), e1 = __expf(w[1] - m);
  w1 = e0 / (e0 + e1);
  w2 = e1 / (e0 + e1);
}

// Per-wave LDS: the head's bias table, the table-gradient accumulator, the per-query softmax statistics and the tokens' regions.
constexpr int NACC = 4;   // table-gradient accumulators per wave (backward): see the table section of winattn_bwd_kernel
struct WaveLds {
  float tab[TABP];
  float tacc[NACC][TABP];
  float mx[NT], inv[NT], sp[NT];
  int regs[NT];
};

// Orientation 1 (S^T: register <-> key, lane <-> query of tile qt): raw products -> biased (masked) scores in place; returns the
// query's row maximum.  The softmax itself is never stored: exp(s - mx) is evaluated once for the row sums and once more where the
// probability is consumed (32 more v_exp_f32 per tile against 32 more live registers: the backward drops from 199 to < 168 VGPRs,
// i.e. from two to three waves per SIMD).
__device__ __forceinline__ float bias_scores(const UirWinAttnArgs& a, const WinGeo& g, const WaveLds& L, int qt, f32x16& s0, f32x16& s1) {
  const int yi = qt * 4 + (g.l32 >> 3), xi = g.l32 & 7;
  const float* tb = L.tab + ((yi + 7) * RP + xi + 7 - g.half * 4);
  const int* rg = L.regs + g.half * 4;
  const int mine = qt ? g.region_q[1] : g.region_q[0];
#pragma unroll
  for (int kt = 0; kt < 2; ++kt)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      f32x16& s = kt ? s1 : s0;
      s[r] = fmaf(s[r], a.scale, tb[-((kt * 4 + r / 4) * RP + (r % 4))]);
    }
  if (a.shift > 0) {                 // one wave-uniform branch around the whole mask pass (a branch per element costs 32 basic blocks)
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        f32x16& s = kt ? s1 : s0;
        s[r] += rg[kt * 32 + (r / 4) * 8 + (r % 4)] != mine ? -100.0f : 0.0f;
      }
  }
  float mx = -__builtin_huge_valf();
#pragma unroll
  for (int r = 0; r < 16; ++r) mx = fmaxf(mx, fmaxf(s0[r], s1[r]));
  return swap32_max(mx);
}

template <int HD>
__global__ __launch_bounds__(256, 2) void winattn_fwd_kernel(const UirWinAttnArgs a, int items) {
  using T = WA<HD>;
  __shared__ WaveLds lds[4];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int item = blockIdx.x * 4 + wave;
  if (item >= items) return;                          // wave-uniform; the kernel has no workgroup barrier
  const int win = item / a.heads, h = item - win * a.heads;
  WaveLds& L = lds[wave];
  WinGeo g;
  make_geo(a, win, lane, g, L.regs);
  for (int t = lane; t < NTAB; t += 64) L.tab[t] = a.table[t * a.heads + h];
  float w1, w2;
  mix_weights(a.w, w1, w2);
  lds_wait();
#pragma unroll 1
  for (int qt = 0; qt < 2; ++qt) {
    f32x16 s0, s1;
    score_pair<HD>(a.K, a.ldk, a.Q, a.ldq, qt, g, h, s0, s1);
    phase();
    const float mx = bias_scores(a, g, L, qt, s0, s1);
    float sum = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) sum += __expf(s0[r] - mx) + __expf(s1[r] - mx);
    const float inv = w1 / swap32_sum(sum);
#pragma unroll
    for (int r = 0; r < 16; ++r) {                    // the mixed attention (:361-367), in place
      const float r0 = fmaxf(s0[r], 0.f), r1 = fmaxf(s1[r], 0.f);
      s0[r] = fmaf(inv, __expf(s0[r] - mx), w2 * r0 * r0);
      s1[r] = fmaf(inv, __expf(s1[r] - mx), w2 * r1 * r1);
    }
    phase();
    int nct = T::NCT;
    asm volatile("" : "+s"(nct));
#pragma unroll 1
    for (int ct = 0; ct < nct; ++ct) {
      f32x16 o;
      acc_times<HD, true>(s0, s1, a.V, a.ldv, g, h, ct, o);
      store_rows<HD>(a.O, a.ldo, g, h, ct, qt, o, 1.0f);
    }
  }
}

template <int HD>
__global__ __launch_bounds__(256, 2) void winattn_bwd_kernel(const UirWinAttnArgs a, int items) {
  using T = WA<HD>;
  __shared__ WaveLds lds[4];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int item = blockIdx.x * 4 + wave;
  if (item >= items) return;
  const int win = item / a.heads, h = item - win * a.heads;
  WaveLds& L = lds[wave];
  WinGeo g;
  make_geo(a, win, lane, g, L.regs);
  for (int t = lane; t < TABP; t += 64) {
    L.tab[t] = t < NTAB ? a.table[t * a.heads + h] : 0.f;
#pragma unroll
    for (int e = 0; e < NACC; ++e) L.tacc[e][t] = 0.f;
  }
  float w1, w2;
  mix_weights(a.w, w1, w2);
  lds_wait();
  float dw1 = 0.f, dw2 = 0.f;

  // ---- orientation 1: register <-> key, lane <-> query.  dQ, the softmax statistics, the table / mixing-weight partials ----
#pragma unroll 1
  for (int qt = 0; qt < 2; ++qt) {
    // dP tiles first: while the score tiles accumulate only 32 + 32 accumulator registers and one chunk of operands are live
    f32x16 s0, s1, d0, d1;
    score_pair<HD>(a.V, a.ldv, a.dO, a.ldo, qt, g, h, d0, d1);          // dP^T[key][query] = v_key . dO_query
    phase();
    score_pair<HD>(a.K, a.ldk, a.Q, a.ldq, qt, g, h, s0, s1);
    phase();
    const float mx = bias_scores(a, g, L, qt, s0, s1);
    float sum = 0.f, sp = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float e0 = __expf(s0[r] - mx), e1 = __expf(s1[r] - mx);
      const float r0 = fmaxf(s0[r], 0.f), r1 = fmaxf(s1[r], 0.f);
      sum += e0 + e1;
      sp = fmaf(d0[r], e0, sp);
      sp = fmaf(d1[r], e1, sp);
      dw2 = fmaf(d0[r], r0 * r0, dw2);
      dw2 = fmaf(d1[r], r1 * r1, dw2);
    }
    const float inv = 1.0f / swap32_sum(sum);
    sp = swap32_sum(sp) * inv;                                           // sum_j dP_ij p_ij
    dw1 += g.half == 0 ? sp : 0.f;                                       // once per query
    if (g.half == 0) {
      const int i = qt * 32 + g.l32;
      L.mx[i] = mx;
      L.inv[i] = inv;
      L.sp[i] = sp;
    }
    const float w1i = w1 * inv;
#pragma unroll
    for (int r = 0; r < 16; ++r) {                                        // softmax backward + relu^2 backward: d0 / d1 become dS^T
      const float r0 = fmaxf(s0[r], 0.f), r1 = fmaxf(s1[r], 0.f);
      d0[r] = fmaf(w1i * __expf(s0[r] - mx), d0[r] - sp, w2 * d0[r] * 2.f * r0);
      d1[r] = fmaf(w1i * __expf(s1[r] - mx), d1[r] - sp, w2 * d1[r] * 2.f * r1);
    }
    {  // relative-position-bias table: entry (yi - yj + 7) * 15 + (xi - xj + 7) collects dS over the pairs with that offset.  Inside one
       // instruction the 32 active lanes are 32 different queries against one key: distinct entries, no collision, fixed order - so a
       // plain read / add / write is enough, and LDS instructions of a wave execute in order.  (As ds_add_f32 the 128 adds per item
       // held the CU's LDS pipe ~90 cycles each: 78 % busy, 39 % of the wave-cycles waiting on it - tools/pmc_winattn.sh.)  The four
       // registers of a key row go to FOUR accumulator tables: four independent round trips per wait instead of one.
      const int yi = qt * 4 + (g.l32 >> 3), xi = g.l32 & 7;
      const int tbo = (yi + 7) * RP + xi + 7 - g.half * 4;
#pragma unroll
      for (int hh = 0; hh < 2; ++hh) {
        if (g.half == hh) {
#pragma unroll
          for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int rq = 0; rq < 4; ++rq) {
              float t[NACC];
#pragma unroll
              for (int e = 0; e < NACC; ++e) t[e] = L.tacc[e][tbo - ((kt * 4 + rq) * RP + e)];
#pragma unroll
              for (int e = 0; e < NACC; ++e) L.tacc[e][tbo - ((kt * 4 + rq) * RP + e)] = t[e] + (kt ? d1[4 * rq + e] : d0[4 * rq + e]);
              phase();                                                    // the next group's reads stay behind these writes
            }
        }
        lds_wait();
      }
    }
    phase();
    int nct = T::NCT;
    asm volatile("" : "+s"(nct));
#pragma unroll 1
    for (int ct = 0; ct < nct; ++ct) {
      f32x16 o;
      acc_times<HD>(d0, d1, a.K, a.ldk, g, h, ct, o);                   // dQ_i = scale * sum_j dS_ij k_j
      store_rows<HD>(a.dQ, a.lddq, g, h, ct, qt, o, a.scale);
    }
  }
  dw1 = uir_wave_sum(dw1);
  dw2 = uir_wave_sum(dw2);
  if (lane == 0) {
    float* o = a.dw_slab + ((long)win * a.heads + h) * 2;
    o[0] = dw1;
    o[1] = dw2;
  }
  lds_wait();
  {
    float* tb = a.dtable_slab + (long)win * (NTAB * a.heads);
    for (int t = lane; t < NTAB; t += 64) tb[t * a.heads + h] = (L.tacc[0][t] + L.tacc[1][t]) + (L.tacc[2][t] + L.tacc[3][t]);
  }

  // ---- orientation 2: register <-> query, lane <-> key of tile kt.  dK_j = scale * sum_i dS_ij q_i,  dV_j = sum_i P_ij dO_i ----
#pragma unroll 1
  for (int kt = 0; kt < 2; ++kt) {
    f32x16 s0, s1, d0, d1;                                                // query tiles 0 / 1 against key tile kt
    score_pair<HD>(a.dO, a.ldo, a.V, a.ldv, kt, g, h, d0, d1);
    phase();
    score_pair<HD>(a.Q, a.ldq, a.K, a.ldk, kt, g, h, s0, s1);
    phase();
    const int yj = kt * 4 + (g.l32 >> 3), xj = g.l32 & 7;
    const float* tb = L.tab + ((7 - yj) * RP + 7 - xj + g.half * 4);
    const int* rg = L.regs + g.half * 4;
    const float *smx = L.mx + g.half * 4, *sinv = L.inv + g.half * 4, *ssp = L.sp + g.half * 4;
    const int mine = kt ? g.region_q[1] : g.region_q[0];
#pragma unroll
    for (int qt = 0; qt < 2; ++qt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        f32x16& s = qt ? s1 : s0;
        s[r] = fmaf(s[r], a.scale, tb[(qt * 4 + r / 4) * RP + (r % 4)]);
      }
    if (a.shift > 0) {
#pragma unroll
      for (int qt = 0; qt < 2; ++qt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          f32x16& s = qt ? s1 : s0;
          s[r] += rg[qt * 32 + (r / 4) * 8 + (r % 4)] != mine ? -100.0f : 0.0f;
        }
    }
#pragma unroll
    for (int qt = 0; qt < 2; ++qt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        f32x16& s = qt ? s1 : s0;
        f32x16& d = qt ? d1 : d0;
        const int i = qt * 32 + (r / 4) * 8 + (r % 4);                    // + half * 4 through the pointers
        const float v = s[r];
        const float p = __expf(v - smx[i]) * sinv[i];
        const float rl = fmaxf(v, 0.f);
        const float dp = d[r];
        s[r] = fmaf(w1, p, w2 * rl * rl);                                  // P
        d[r] = fmaf(w1 * p, dp - ssp[i], w2 * dp * 2.f * rl);              // dS
      }
    phase();
    int nct = T::NCT;
    asm volatile("" : "+s"(nct));
#pragma unroll 1
    for (int ct = 0; ct < nct; ++ct) {
      f32x16 o;
      acc_times<HD>(d0, d1, a.Q, a.ldq, g, h, ct, o);
      store_rows<HD>(a.dK, a.lddk, g, h, ct, kt, o, a.scale);
      phase();
      acc_times<HD>(s0, s1, a.dO, a.ldo, g, h, ct, o);
      store_rows<HD>(a.dV, a.lddv, g, h, ct, kt, o, 1.0f);
    }
  }
}

// d(self.w): (w1, w2) = softmax(w), dsum = sum over (window, head) partials already reduced into g[2]
__global__ void winattn_wgrad_kernel(const float* __restrict__ w, const float* __restrict__ g, float* __restrict__ dw) {
  if (threadIdx.x == 0) {
    float w1, w2;
    mix_weights(w, w1, w2);
    const float dot = w1 * g[0] + w2 * g[1];
    dw[0] = w1 * (g[0] - dot);
    dw[1] = w2 * (g[1] - dot);
  }
}

int check(const UirWinAttnArgs& a, bool bwd) {
  UIR_REQUIRE(a.Q && a.K && a.V && a.table && a.w && a.B > 0 && a.heads > 0 && a.C > 0 && a.C % a.heads == 0);
  UIR_REQUIRE(a.ws == WS && a.H % WS == 0 && a.W % WS == 0 && a.shift >= 0 && a.shift < WS);
  const int hd = a.C / a.heads;
  UIR_REQUIRE(hd == 8 || hd == 16 || hd == 32 || hd == 64 || hd == 128);
  UIR_REQUIRE(a.ldq >= a.C && a.ldk >= a.C && a.ldv >= a.C && ((a.ldq | a.ldk | a.ldv) & 3) == 0);
  UIR_REQUIRE(uir_aligned16(a.Q) && uir_aligned16(a.K) && uir_aligned16(a.V));
  const long px = (long)a.B * a.H * a.W;               // element offsets are 32-bit
  int ldmax = a.ldq > a.ldk ? a.ldq : a.ldk;
  ldmax = ldmax > a.ldv ? ldmax : a.ldv;
  if (!bwd) {
    UIR_REQUIRE(a.O && a.ldo >= a.C && (a.ldo & 3) == 0 && uir_aligned16(a.O));
    ldmax = ldmax > a.ldo ? ldmax : a.ldo;
  } else {
    UIR_REQUIRE(a.dO && a.dQ && a.dK && a.dV && a.dtable_slab && a.dw_slab && a.ldo >= a.C && (a.ldo & 3) == 0);
    UIR_REQUIRE(a.lddq >= a.C && a.lddk >= a.C && a.lddv >= a.C && ((a.lddq | a.lddk | a.lddv) & 3) == 0);
    UIR_REQUIRE(uir_aligned16(a.dO) && uir_aligned16(a.dQ) && uir_aligned16(a.dK) && uir_aligned16(a.dV));
    const int m1 = a.lddq > a.lddk ? a.lddq : a.lddk, m2 = a.lddv > a.ldo ? a.lddv : a.ldo;
    ldmax = ldmax > m1 ? ldmax : m1;
    ldmax = ldmax > m2 ? ldmax : m2;
  }
  UIR_REQUIRE(px * ldmax < (1L << 31));
  return UIR_OK;
}

}  // namespace

extern "C" int uir_winattn_fwd(const UirWinAttnArgs* p, void* stream) {
  const UirWinAttnArgs& a = *p;
  if (int rc = check(a, false)) return rc;
  const int items = a.B * (a.H / WS) * (a.W / WS) * a.heads;
  const dim3 grid((items + 3) / 4);
  hipStream_t st = static_cast<hipStream_t>(stream);
  switch (a.C / a.heads) {
    case 8: hipLaunchKernelGGL(winattn_fwd_kernel<8>, grid, dim3(256), 0, st, a, items); break;
    case 16: hipLaunchKernelGGL(winattn_fwd_kernel<16>, grid, dim3(256), 0, st, a, items); break;
    case 32: hipLaunchKernelGGL(winattn_fwd_kernel<32>, grid, dim3(256), 0, st, a, items); break;
    case 64: hipLaunchKernelGGL(winattn_fwd_kernel<64>, grid, dim3(256), 0, st, a, items); break;
    default: hipLaunchKernelGGL(winattn_fwd_kernel<128>, grid, dim3(256), 0, st, a, items); break;
  }
  return uir_launch_status();
}

extern "C" int uir_winattn_bwd(const UirWinAttnArgs* p, void* stream) {
  const UirWinAttnArgs& a = *p;
  if (int rc = check(a, true)) return rc;
  const int items = a.B * (a.H / WS) * (a.W / WS) * a.heads;
  const dim3 grid((items + 3) / 4);
  hipStream_t st = static_cast<hipStream_t>(stream);
  switch (a.C / a.heads) {
    case 8: hipLaunchKernelGGL(winattn_bwd_kernel<8>, grid, dim3(256), 0, st, a, items); break;
    case 16: hipLaunchKernelGGL(winattn_bwd_kernel<16>, grid, dim3(256), 0, st, a, items); break;
    case 32: hipLaunchKernelGGL(winattn_bwd_kernel<32>, grid, dim3(256), 0, st, a, items); break;
    case 64: hipLaunchKernelGGL(winattn_bwd_kernel<64>, grid, dim3(256), 0, st, a, items); break;
    default: hipLaunchKernelGGL(winattn_bwd_kernel<128>, grid, dim3(256), 0, st, a, items); break;
  }
  return uir_launch_status();
}

extern "C" int uir_winattn_wgrad(const float* w, const float* dsum, float* dw, void* stream) {
  UIR_REQUIRE(w && dsum && dw);
  hipLaunchKernelGGL(winattn_wgrad_kernel, dim3(1), dim3(64), 0, static_cast<hipStream_t>(stream), w, dsum, dw);
  return uir_launch_status();
}
