// uir_dwln_bwd: the backward of  LayerNorm -> 1x1 conv (C -> CH) -> depthwise 3x3  in ONE pass over the pixels:
//
//   dP  = dw3x3^T(G)                                  gradient of the depthwise INPUT - never leaves the chip
//   dT  = dP . W1,   dX = dRes + LayerNorm'(dT)       the block's input gradient
//   dWdw[c][t] = sum_q XP[q][c] G[q - off_t][c]       depthwise weight gradient
//   dW1[c][n]  = sum_q dP[q][c] (xhat[q][n] gamma[n] + beta[n])     1x1 weight gradient
//   dgamma[n]  = sum_q dT[q][n] xhat[q][n],  dbeta[n] = sum_q dT[q][n]
//
// It is used twice per TransformerBlock of SpectralTransformer.py:142-148: for norm2 -> ffn.project_in -> ffn.conv (:122-124,
// "split" column layout of the GDFN tensors: x1 half at [0, h), x2 half at [hp, hp + h)) and for norm1 -> attn.qkv -> attn.qkv_conv
// (:81-82,94; 3C consecutive columns).  It replaces three launches per use - uir_dwconv_bwd_fused (dP written to HBM), the
// LayerNorm-epilogue uir_gemm (dP read) and the uir_gram of the 1x1 weight gradient (dP read again): 1088 -> ~560 floats per pixel
// at C = 32 for the GDFN half, 608 -> ~320 for the qkv half.
//
// One persistent workgroup walks 8 x 16 pixel tiles.  Per tile the LayerNorm input is normalised into LDS (xhat); then the CH
// channels are walked in chunks of 32 (16 + 16 in the split layout):
//   (a) G chunk on tile + ring: global -> registers (one chunk ahead) -> LDS
//   (b) transposed depthwise 3x3 from LDS with the rotating three-row register window of uir_gdfn_fwd (packed f32 over the
//       channel pair); the SAME window elements times the centre XP value give the 9 depthwise weight-gradient terms;
//       dP chunk -> LDS; the weight-gradient terms are summed over the wave's four strips by shuffles, over waves through LDS
//   (a') XP = NULL (round 5, what the train step passes): the centre XP values of the walk are RECOMPUTED - one 32-channel x 32-pixel tile
//       per wave (its own walk pixels) = W1 chunk . LayerNorm(xhat), split-bf16 MFMA, parked in the dP buffer at the element the same thread
//       overwrites in (b); the forward then never writes the pre-depthwise tensor
//   (c) dT += dP chunk . W1 chunk        one 32-pixel row tile per wave, accumulators live across chunks; split-bf16 MFMA in the XP = NULL
//       variants (an f32 MFMA does not run beside the SIMD's VALU and this kernel is bound by the sum of the two), v_mfma_f32_32x32x2_f32 otherwise
//   (d) dW1 chunk += dP^T . (LN output)  v_mfma_f32_16x16x4_f32: each wave owns one 16 x 16 granule of the chunk's 32 x C tile and
//       contracts over the tile's pixels, so a wave keeps 4 accumulator registers per chunk across all its tiles
// and the tile ends with the LayerNorm backward on the dT accumulators (channel on the lane: row means by xor-shuffles).
// Partial sums leave as one slab row per workgroup (fixed summation order everywhere: bitwise run-to-run).
#include "uir_common.h"
#include <type_traits>
#include "../../include/uir_kernels.h"
#include "uir_stamps.h"

namespace {

typedef float f32x4v __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void lds_barrier2() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
// uniform base pointer + 32-bit byte offset: hipcc emits the scalar-base form `global_load_dword v, v_off, s[base:base+1]` (one offset
// VGPR, no 64-bit address arithmetic per load); `base[unsigned_index]` made it build a 64-bit VGPR address for every load.
__device__ __forceinline__ float ldg_off(const float* base, unsigned byte_off) {
  return *reinterpret_cast<const float*>(reinterpret_cast<const char*>(base) + byte_off);
}
__device__ __forceinline__ f32x4v mfma16(float a, float b, f32x4v c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

// Cross-lane sums without the LDS pipe (__shfl_xor lowers to ds_bpermute_b32: an address VGPR, an LDS instruction and its latency
// per step).  Inside a 16-lane row: DPP butterfly (quad_perm [1,0,3,2], [2,3,0,1], row_half_mirror, row_mirror: every lane ends with
// the row sum).  Across rows: gfx950's v_permlane16_swap / v_permlane32_swap - called with both operands = x they return
// {x.r0, x.r0, x.r2, x.r2} | {x.r1, x.r1, x.r3, x.r3} resp. {x.lo32, x.lo32} | {x.hi32, x.hi32}; their sum is the lane-wise total.
template <int CTRL>
__device__ __forceinline__ float dpp_add(float x) {
  return x + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, 0xF, 0xF, false));
}
__device__ __forceinline__ float row_sum16(float x) {
  x = dpp_add<0xB1>(x);
  x = dpp_add<0x4E>(x);
  x = dpp_add<0x141>(x);
  return dpp_add<0x140>(x);
}
__device__ __forceinline__ float swap16_sum(float x) {        // lanes l and l ^ 16
  const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float swap32_sum(float x) {        // lanes l and l ^ 32
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

template <int C, int NCH, bool SPLIT>
struct DwLn {
  static constexpr int TH = 8, TW = 16, RW = TW + 2, RH = TH + 2;
  static constexpr int NPX1 = RW * RH;                    // 180 tile + ring pixels
  static constexpr int NI = TH * TW;                      // 128 interior pixels = 4 MFMA row tiles, one per wave
  static constexpr int NLD = (NPX1 * 8 + 255) / 256;      // float4 loads per thread and G chunk (6)
  static constexpr int S1 = 40, SG = 36;                  // LDS row strides (floats): G chunk / dP chunk
  static constexpr int SXT = NI + 4;                      // xhat is kept TRANSPOSED, [channel][pixel]: (d) reads four consecutive pixels of one
                                                          // channel with one ds_read_b128 (four MFMA steps), the epilogue four of its 16 rows
  static constexpr int GS1_ROWS = NLD * 32;               // 192 (rows >= NPX1 hold zeros)
  static constexpr int NXL = (NI * (C / 4)) / 256;        // float4 loads per thread of the x tile (4 at C = 32, 2 at C = 16)
  static constexpr int NG = 2 * (C / 16);                 // 16 x 16 granules of a chunk's dW1 tile (4 at C = 32, 2 at C = 16)
  static constexpr int PARTS = 4 / NG;                    // waves sharing a granule split the pixels (C = 16: 2)
  static constexpr int STEPS = NI / 4 / PARTS;            // 16x16x4 MFMA steps per wave and chunk
  static constexpr int OFF_GS = GS1_ROWS * S1;
  static constexpr int OFF_XH = OFF_GS + NI * SG;
  static constexpr int OFF_RS = OFF_XH + NI * SG;          // (the [32][NI + 4] transposed xhat fits the [NI][36] it replaced)
  static_assert(32 * SXT <= NI * SG, "xhat buffer");
  static constexpr int OFF_WR = OFF_RS + NI;
  static constexpr int SW = C + 4;                         // row stride of the staged W1 chunk [32 rows][C] (16-byte rows)
  static constexpr int OFF_W1 = OFF_WR + 4 * 16 * 18;
  static constexpr int SWF = 20;                           // row stride of the staged depthwise taps [16 channel pairs][9 A | 9 B] (16-byte rows)
  static constexpr int OFF_WF = OFF_W1 + 32 * SW;
  static constexpr int OFF_GB = OFF_WF + 16 * SWF;         // recompute mode: (gamma, beta) pairs of the LayerNorm [C][2]
  static constexpr int LDS_FLOATS = OFF_GB + 2 * C;
  static_assert(C == 16 || C == 32, "channel count");
};

// Per-thread coordinates are RE-DERIVED from an opaque copy of threadIdx.x at the top of every scope that uses them (tile, chunk,
// epilogue).  Derived once at kernel scope, hipcc hoisted every address / index expression built from them out of both loops
// (~150 loop-invariant VGPRs), spilled ~100 of them and reloaded some INSIDE the chunk loop - and a scratch reload is a VMEM
// load: its s_waitcnt vmcnt(0) drains every prefetch in flight.  Recomputing ~60 integer ops per chunk is far cheaper.
#define UIR_DWLN_COORDS                                                                                                   \
  int t_ = threadIdx.x;                                                                                                   \
  asm volatile("" : "+v"(t_));                                                                                            \
  const int lane = t_ & 63, wave = __builtin_amdgcn_readfirstlane(t_ >> 6), li = lane & 31, h2 = lane >> 5;              \
  const int cp = lane >> 2, sx = wave * 4 + (lane & 3);    /* a wave's four strips of a channel pair = one lane quad */        \
  const int l16 = lane & 15, j16 = lane >> 4;                                                                             \
  const int gr_k = (T::NG == 4) ? (wave >> 1) : (wave & 1), gr_n = (T::NG == 4) ? (wave & 1) : 0;                         \
  const int part = (T::NG == 4) ? 0 : (wave >> 1), n16 = 16 * gr_n + l16;                                                 \
  (void)li; (void)h2; (void)cp; (void)sx; (void)l16; (void)j16; (void)gr_k; (void)part; (void)n16

// RC (a.XP == NULL): the centre pre-conv values XP = LayerNorm(x) . W1^T are RECOMPUTED per chunk from the normalised tile and the staged W1
// chunk (f32 MFMA, one 32-channel x 32-pixel tile per wave = exactly the pixels of the wave's four walk columns), parked in the dP buffer at
// the element the same thread overwrites with dP in the walk.  The forward then never writes the pre-depthwise tensor (96 resp. 192 floats
// per pixel at C = 32: 43 % of uir_qkv_fwd's / uir_gdfn_fwd's bytes) and this kernel does not read it.
template <int C, int NCH, bool SPLIT, bool RC>
__global__ __launch_bounds__(256, 2) void uir_dwln_bwd_kernel(const UirDwLnBwdArgs a) {
  using T = DwLn<C, NCH, SPLIT>;
  extern __shared__ float smem[];
  float* GS1 = smem;
  float* Gs = smem + T::OFF_GS;
  float* XH = smem + T::OFF_XH;
  float* RS = smem + T::OFF_RS;
  float* WR = smem + T::OFF_WR;
  float* W1s = smem + T::OFF_W1;
  float* WFs = smem + T::OFF_WF;
  float* GB = smem + T::OFF_GB;
  if (RC && threadIdx.x < C) {                               // visible after the first tile's barriers
    GB[2 * threadIdx.x] = a.gamma[threadIdx.x];
    GB[2 * threadIdx.x + 1] = a.beta[threadIdx.x];
  }
  const int tiles_x = (a.Wd + T::TW - 1) / T::TW, tiles_y = (a.H + T::TH - 1) / T::TH;
  const int ntiles = a.B * tiles_x * tiles_y;
  const int h = a.h, hp = a.hp, ch = a.ch;
  const int chcols = SPLIT ? 2 * hp : ch;                 // columns of G / XP that exist

  // chunk geometry: two groups of 16 channels; column of the group in G / XP, real weight row, validity of channel i of the group
  // a.il (GDFN layout): the chunk's 16 + 16 columns are consecutive in memory (UirGdfnFwdArgs.il) - only the COLUMNS change, the
  // weight rows and the validity of a channel stay those of the split layout
  const bool il = SPLIT && a.il != 0;
  auto colA = [&](int ci) { return (SPLIT && !il) ? 16 * ci : 32 * ci; };
  auto colB = [&](int ci) { return (SPLIT && !il) ? hp + 16 * ci : 32 * ci + 16; };
  auto rowA = [&](int ci, int i) { return (SPLIT ? 16 * ci : 32 * ci) + i; };
  auto rowB = [&](int ci, int i) { return SPLIT ? h + 16 * ci + i : 32 * ci + 16 + i; };
  auto okA = [&](int ci, int i) { return SPLIT ? 16 * ci + i < h : 32 * ci + i < ch; };
  auto okB = [&](int ci, int i) { return SPLIT ? 16 * ci + i < h : 32 * ci + 16 + i < ch; };
  auto g_col = [&](int ci, int e) {                        // first column of float4 slot e & 7 of a G / XP chunk row
    const int j = e & 7;
    return j < 4 ? colA(ci) + 4 * j : colB(ci) + 4 * (j - 4);
  };

  float gam16, bet16, gam_n;
  {
    UIR_DWLN_COORDS;
    gam16 = a.gamma[n16];
    bet16 = a.beta[n16];
    gam_n = li < C ? a.gamma[li] : 0.f;
  }
  // dW1 granule accumulators, one per chunk, live across all tiles.  Named registers selected by a uniform branch: the chunk loop
  // stays ROLLED (unrolled, hipcc hoisted every chunk's weight / address work and spilled 339 VGPRs).
  static_assert(NCH <= 6, "accumulator set");
  const f32x4v zero4 = f32x4v{0.f, 0.f, 0.f, 0.f};
  f32x4v dw0 = zero4, dw1 = zero4, dw2 = zero4, dw3 = zero4, dw4 = zero4, dw5 = zero4;
#define DW_GET(ci, v) do { if (ci == 0) v = dw0; else if (ci == 1) v = dw1; else if (ci == 2) v = dw2; else if (ci == 3) v = dw3; else if (ci == 4) v = dw4; else v = dw5; } while (0)
#define DW_PUT(ci, v) do { if (ci == 0) dw0 = v; else if (ci == 1) dw1 = v; else if (ci == 2) dw2 = v; else if (ci == 3) dw3 = v; else if (ci == 4) dw4 = v; else dw5 = v; } while (0)
  float dgam = 0.f, dbet = 0.f;
  // depthwise weight-gradient sums of the workgroup: entry e (< 288 = 16 channels x 18) of chunk ci lives in thread e (da*) resp.
  // thread e - 256 (db*, threads 0..31) - in registers, not in LDS (the 6.9 KB they took hold the staged W1 chunk now)
  float da0 = 0.f, da1 = 0.f, da2 = 0.f, da3 = 0.f, da4 = 0.f, da5 = 0.f, db0 = 0.f, db1 = 0.f, db2 = 0.f, db3 = 0.f, db4 = 0.f, db5 = 0.f;
#define DA_ADD(ci, va, vb) do { if (ci == 0) { da0 += va; db0 += vb; } else if (ci == 1) { da1 += va; db1 += vb; } else if (ci == 2) { da2 += va; db2 += vb; } \
                                else if (ci == 3) { da3 += va; db3 += vb; } else if (ci == 4) { da4 += va; db4 += vb; } else { da5 += va; db5 += vb; } } while (0)
#define DA_GET(ci, va, vb) do { if (ci == 0) { va = da0; vb = db0; } else if (ci == 1) { va = da1; vb = db1; } else if (ci == 2) { va = da2; vb = db2; } \
                                else if (ci == 3) { va = da3; vb = db3; } else if (ci == 4) { va = da4; vb = db4; } else { va = da5; vb = db5; } } while (0)

  // tile coordinates: first row / column, first pixel row of the image, rows of the tile inside the image (>= 1)
  struct Tc { int y0, x0; unsigned img; int nrows; };
  auto coords = [&](int tile) {
    int bid = tile < ntiles ? tile : ntiles - 1;             // past the end: a valid tile (its prefetch is never consumed)
    const int tx = bid % tiles_x;
    bid /= tiles_x;
    const int ty = bid % tiles_y;
    const int b = bid / tiles_y;
    Tc c;
    c.y0 = ty * T::TH; c.x0 = tx * T::TW;
    c.img = (unsigned)b * (unsigned)(a.H * a.Wd);            // 32-bit row indices / offsets everywhere (checked by the launcher)
    c.nrows = a.H - c.y0;
    return c;
  };
  const unsigned xrstep = (unsigned)(a.Wd * a.ldp);
  // ring-pixel row of G load slot e = t + 256 it: byte offset of the row in G, or ~0u outside the image / past the ring
  auto g_off = [&](int e, const Tc& c) -> unsigned {
    const int p = e >> 3;
    const int ry = p / T::RW, rx = p - ry * T::RW;
    const int iy = c.y0 + ry - 1, ix = c.x0 + rx - 1;
    const bool ok = p < T::NPX1 && iy >= 0 && iy < a.H && ix >= 0 && ix < a.Wd;
    return ok ? (c.img + (unsigned)(iy * a.Wd + ix)) * (unsigned)a.ldg * 4u : ~0u;
  };
  // Prefetches carry RAW values from clamped (always valid) addresses; the zeroing of out-of-image pixels / missing columns is a
  // multiplication where the value is consumed.  (`ok ? load : 0` becomes a branch around the load plus an immediate
  // s_waitcnt: no prefetch at all; and vmcnt retires in issue order, so whatever is needed first must be issued first.)
  // gv / goff / xpv / wf_n always describe the NEXT (tile, chunk) unit: chunk ci + 1 of the tile, or chunk 0 of the workgroup's
  // next tile - a tile starts with its first chunk already in flight.
  float4 gv[T::NLD];
  unsigned goff[T::NLD];                                     // per tile (6 registers across the chunk loop: ~280 VALU ops per chunk less)
  auto load_g = [&](int ci, int t) {
#pragma unroll
    for (int it = 0; it < T::NLD; ++it) {
      const int col = g_col(ci, t + 256 * it);
      const unsigned off = (goff[it] != ~0u ? goff[it] : 0u) + (unsigned)(col + 4 <= chcols ? col : 0) * 4u;
      gv[it] = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(a.G) + off);
    }
  };
  // centre XP values of the thread's walk: pixel (row y, column sx), channels cp of group A and group B
  f32x2 xpv[T::TH];
  auto load_xp = [&](int ci, int cp_, int sx_, const Tc& c) {
    const bool colok = c.x0 + sx_ < a.Wd;
    const unsigned xrow0 = (c.img + (unsigned)(c.y0 * a.Wd + (colok ? c.x0 + sx_ : 0))) * (unsigned)a.ldp;
    const int ca = colA(ci) + cp_, cb = colB(ci) + cp_;
    const unsigned oa = (xrow0 + (unsigned)(ca < chcols ? ca : 0)) * 4u, ob = (xrow0 + (unsigned)(cb < chcols ? cb : 0)) * 4u;
#pragma unroll
    for (int y = 0; y < T::TH; ++y) {
      const float* rowp = a.XP + (size_t)((unsigned)(y < c.nrows ? y : 0) * xrstep);   // uniform: scalar base + one offset VGPR
      xpv[y] = f32x2{ldg_off(rowp, oa), ldg_off(rowp, ob)};
    }
  };
  // depthwise taps of the chunk's 32 channels (288 floats): element t of the chunk by thread t (and 256 + t by threads 0..31), a chunk
  // ahead, staged through LDS with the G rows like the W1 chunk - they were 18 global dword loads per lane and chunk (a channel
  // pair's 2 x 9 taps, replicated over the pair's 4 strips), the other half of the small loads behind the `xp/wf-issue` stamp.
  // LDS row of channel pair cp: window order (position t meets tap 8 - t), 9 of group A then 9 of group B; missing channels hold 0
  f32x2 wf[9];
  float wfpre0, wfpre1;
  auto wf_src = [&](int ci, int e, bool& ok) -> unsigned {
    const int r = e / 9, tap = e - 9 * r, i = r & 15;
    ok = r < 16 ? okA(ci, i) : okB(ci, i);
    return (unsigned)((ok ? (r < 16 ? rowA(ci, i) : rowB(ci, i)) : 0) * 9 + tap);
  };
  auto load_wf = [&](int ci, int t) {
    bool ok;
    wfpre0 = a.Wdw[wf_src(ci, t, ok)];
    wfpre1 = a.Wdw[wf_src(ci, 256 + (t & 31), ok)];
  };
  auto store_wf = [&](int ci, int t) {
    bool ok0, ok1;
    (void)wf_src(ci, t, ok0);
    (void)wf_src(ci, 256 + (t & 31), ok1);
    {
      const int r = t / 9, tap = t - 9 * r;
      WFs[(r & 15) * T::SWF + (r >> 4) * 9 + (8 - tap)] = ok0 ? wfpre0 : 0.f;
    }
    if (t < 32) {
      const int e = 256 + t, r = e / 9, tap = e - 9 * r;
      WFs[(r & 15) * T::SWF + (r >> 4) * 9 + (8 - tap)] = ok1 ? wfpre1 : 0.f;
    }
  };
  // the chunk's 32 rows of W1 (16 of group A, 16 of group B): ONE float4 per thread, a chunk ahead, staged through LDS with the G rows;
  // every lane then reads its 16 B-operand values with LDS loads - they were 16 global dword loads per lane and chunk, a third of
  // the small loads whose issue stalls (behind the G stream in the memory pipe) cost the kernel ~20 % (tools/probes/stamps)
  float4 w1pre;
  auto load_w1 = [&](int ci, int t) {
    const int r_ = t / (C / 4), q_ = t - r_ * (C / 4), r = r_ < 32 ? r_ : 31;        // C = 16: threads 128.. repeat row 31 (not stored)
    const int i = r & 15;
    const int base = r < 16 ? (SPLIT ? 16 * ci : 32 * ci) : (SPLIT ? h + 16 * ci : 32 * ci + 16);
    const int limA = SPLIT ? h - 16 * ci : ch - 32 * ci;
    const bool ok = i < (r < 16 || SPLIT ? limA : limA - 16);
    w1pre = *reinterpret_cast<const float4*>(a.W1 + (unsigned)((ok ? base + i : 0) * C + 4 * q_));
  };
  {
    UIR_DWLN_COORDS;
    const Tc c0 = coords(blockIdx.x);
#pragma unroll
    for (int it = 0; it < T::NLD; ++it) goff[it] = g_off(t_ + 256 * it, c0);
    load_wf(0, t_);
    load_w1(0, t_);
    load_g(0, t_);
    if constexpr (!RC) load_xp(0, cp, sx, c0);
  }

  UIR_ST_DECL;
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const Tc tc = coords(tile), tn = coords(tile + gridDim.x);
    const int y0 = tc.y0, x0 = tc.x0, nrows = tc.nrows;
    const unsigned img = tc.img;
    {
      UIR_DWLN_COORDS;
      // ---- xhat of the tile's interior -> LDS (all waves are past the previous tile's reads of XH / RS after this barrier) ----
      // all loads of the tile in flight together, then the LDS stores: as one loop hipcc waited for each (x, stats) pair with
      // vmcnt(0) before the next pair was issued - NXL exposed round trips per tile (tools/scan_isa_chains.py)
      float4 xv[T::NXL];
      float2 sv[T::NXL];
      bool okv[T::NXL];
#pragma unroll
      for (int it = 0; it < T::NXL; ++it) {
        const int e = t_ + 256 * it, p = e / (C / 4), q = e - p * (C / 4);
        const int py = p / T::TW, px = p - py * T::TW;
        const int iy = y0 + py, ix = x0 + px;
        okv[it] = iy < a.H && ix < a.Wd;
        const unsigned m = okv[it] ? img + (unsigned)(iy * a.Wd + ix) : img;
        xv[it] = *reinterpret_cast<const float4*>(a.X + (m * (unsigned)a.ldx + 4u * q));
        sv[it] = *reinterpret_cast<const float2*>(a.stats + m * 2u);
      }
      __builtin_amdgcn_sched_barrier(0);
      lds_barrier2();
#pragma unroll
      for (int it = 0; it < T::NXL; ++it) {
        const int e = t_ + 256 * it, p = e / (C / 4), q = e - p * (C / 4);
        const float4 v = xv[it];
        const float2 st = sv[it];
        const float rs = okv[it] ? st.y : 0.f;
        float* xd = XH + 4 * q * T::SXT + p;
        xd[0] = (v.x - st.x) * rs;
        xd[T::SXT] = (v.y - st.x) * rs;
        xd[2 * T::SXT] = (v.z - st.x) * rs;
        xd[3 * T::SXT] = (v.w - st.x) * rs;
        if (q == 0) RS[p] = rs;
      }
    }
    UIR_ST(0);

    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    UirBf3 bvp[RC ? C / 16 : 1];                             // RC: LayerNorm output of this lane's (pixel, n) slots as bf16 planes, the B operand of the XP recompute

#pragma unroll 1
    for (int ci = 0; ci < NCH; ++ci) {
      UIR_DWLN_COORDS;
      const bool colok = x0 + sx < a.Wd;
      const bool last = ci + 1 == NCH;
      const int cn = last ? 0 : ci + 1;                      // next unit: chunk ci + 1 of this tile, or chunk 0 of the next tile
      // `plain` (uniform): the tile's ring lies inside the image and all 32 channels of the chunk exist - 2/3 of the (tile, chunk) units
      // at 256 x 256, C = 32.  No value needs masking there: the masks below (~90 VALU per chunk) are skipped
      const bool plain = y0 >= 1 && x0 >= 1 && y0 + T::TH + 1 <= a.H && x0 + T::TW + 1 <= a.Wd &&
                         (SPLIT ? 16 * ci + 16 <= h : 32 * ci + 32 <= ch);
      // ---- (a) this chunk's G (tile + ring) -> LDS ----
      if (plain) {
#pragma unroll
        for (int it = 0; it < T::NLD; ++it) {
          const int e = t_ + 256 * it;
          *reinterpret_cast<float4*>(GS1 + (e >> 3) * T::S1 + 4 * (e & 7)) = gv[it];       // (rows >= NPX1: unused filler)
        }
      } else {
#pragma unroll
        for (int it = 0; it < T::NLD; ++it) {
          const int e = t_ + 256 * it;
          const float k = (goff[it] != ~0u && g_col(ci, e) + 4 <= chcols) ? 1.f : 0.f;
          *reinterpret_cast<float4*>(GS1 + (e >> 3) * T::S1 + 4 * (e & 7)) = make_float4(gv[it].x * k, gv[it].y * k, gv[it].z * k, gv[it].w * k);
        }
      }
      store_wf(ci, t_);
      if (t_ / (C / 4) < 32) *reinterpret_cast<float4*>(W1s + (t_ / (C / 4)) * T::SW + 4 * (t_ % (C / 4))) = w1pre;
      lds_barrier2();
      UIR_ST(1);
      f32x16 xd;
      if constexpr (RC) {
        // XP tile of this wave: D[chunk channel r][pixel q] = sum_n W1s[r][n] (xhat[n][pixel q] gamma[n] + beta[n]); MFMA column q = li is the
        // pixel (row li >> 2, column 4 wave + (li & 3)) - the walk's pixels of this wave.  Every wave is past (c) / (d) of the previous chunk (the
        // barrier above): Gs is free
        const int pix = (li >> 2) * T::TW + 4 * wave + (li & 3);
        // split-bf16 (three planes per operand, six products, f32-accurate: uir_common.h): the f32 MFMA keeps the SIMD's VALU from issuing while it
        // runs and this kernel is bound by the sum of the two; the bf16 MFMA runs beside the VALU.  k slot t of step s is n = 16 s + 8 (t / 4) +
        // 4 h2 + t % 4 for both operands: a lane's A values of a step are two float4 of its W1s row
        if (ci == 0) {                                       // the B operand is the tile's, not the chunk's: split once, kept across the chunk loop
#pragma unroll
          for (int s2 = 0; s2 < C / 16; ++s2) {
            float xb[8];
#pragma unroll
            for (int t = 0; t < 8; ++t) {
              const int n = 16 * s2 + 8 * (t >> 2) + 4 * h2 + (t & 3);
              const float2 g2 = *reinterpret_cast<const float2*>(GB + 2 * n);
              xb[t] = fmaf(XH[n * T::SXT + pix], g2.x, g2.y);
            }
            bvp[s2] = uir_split8(xb);
          }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) xd[r] = 0.f;
        const float* ap = W1s + li * T::SW + 4 * h2;
#pragma unroll
        for (int s2 = 0; s2 < C / 16; ++s2) {
          const f32x4v a0 = *reinterpret_cast<const f32x4v*>(ap + 16 * s2), a1 = *reinterpret_cast<const f32x4v*>(ap + 16 * s2 + 8);
          xd = uir_mfma_split(uir_split8(make_float4(a0[0], a0[1], a0[2], a0[3]), make_float4(a1[0], a1[1], a1[2], a1[3])), bvp[s2], xd);
        }
        // (stored to Gs below, behind the weight reads and the next unit's loads: the MFMA chain runs under them)
      }
      // issue order = need order (vmcnt retires in order): this chunk's 1x1 weights now (used after the walk); the next chunk's
      // G, XP and depthwise taps after the walk, when its window / tap / accumulator registers are free - they land during (c), (d)
      // (c) B operand: W1[row of k-slot 8 jj + 4 h2 + e][n = li].  k-slots 0..15 are channels i = 8 (jj & 1) + 4 h2 + e of group A,
      // 16..31 the same i of group B: a uniform row base per group + one per-lane offset + compile-time immediates.  Rows past the
      // real channel count (last chunk only) are re-pointed at row 0: their dP is exactly zero (masked taps), any finite weight does.
      float w1v[16];
      {
        const float* wp = W1s + 4 * h2 * T::SW + (li < C ? li : C - 1);
#pragma unroll
        for (int s = 0; s < 16; ++s) w1v[s] = wp[((s < 8 ? 0 : 16) + 8 * ((s >> 2) & 1) + (s & 3)) * T::SW];
      }
      const f32x2 xm = f32x2{colok && colA(ci) + cp < chcols ? 1.f : 0.f, colok && colB(ci) + cp < chcols ? 1.f : 0.f};
      // the next unit's G rows (24 of the ~40 KB a chunk moves per workgroup) leave NOW, behind w1v in issue order: their registers
      // were freed by the LDS stores above, and they get the walk + (c) + (d) to land instead of (c) + (d) alone - with two workgroups
      // per CU the bytes in flight bound the bandwidth (Little's law), not the issue rate
      if (last) {                                            // (uniform branch, no loads inside: the load count per chunk stays static)
#pragma unroll
        for (int it = 0; it < T::NLD; ++it) goff[it] = g_off(t_ + 256 * it, tn);
      }
      load_w1(cn, t_);
      load_wf(cn, t_);
      load_g(cn, t_);
      {
        // this chunk's taps: 18 consecutive floats of the pair's LDS row (4 x b128 + b64)
        const float* wr = WFs + cp * T::SWF;
        const float4 q0 = *reinterpret_cast<const float4*>(wr), q1 = *reinterpret_cast<const float4*>(wr + 4),
                     q2 = *reinterpret_cast<const float4*>(wr + 8), q3 = *reinterpret_cast<const float4*>(wr + 12);
        const float2 q4 = *reinterpret_cast<const float2*>(wr + 16);
        const float w18[18] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w, q2.x, q2.y, q2.z, q2.w, q3.x, q3.y, q3.z, q3.w, q4.x, q4.y};
#pragma unroll
        for (int t = 0; t < 9; ++t) wf[t] = f32x2{w18[t], w18[9 + t]};
      }
      if constexpr (RC) {
        const int pix = (li >> 2) * T::TW + 4 * wave + (li & 3);
        float* xo = Gs + pix * T::SG + 4 * h2;               // registers 4 j .. 4 j + 3 = channels 8 j + 4 h2 .. + 3 of pixel q
#pragma unroll
        for (int j = 0; j < 4; ++j) *reinterpret_cast<f32x4v*>(xo + 8 * j) = f32x4v{xd[4 * j], xd[4 * j + 1], xd[4 * j + 2], xd[4 * j + 3]};
        // read back in the walk by lanes of THIS wave only (LDS instructions of a wave execute in order)
      }
      UIR_ST(2);

      // ---- (b) transposed depthwise 3x3 + depthwise weight-gradient terms ----
      f32x2 dwa[9];
#pragma unroll
      for (int t = 0; t < 9; ++t) dwa[t] = f32x2{0.f, 0.f};
      auto walk = [&](auto plain_c) {
        constexpr bool PL = decltype(plain_c)::value;
        const float* rp = GS1 + sx * T::S1 + cp;
        f32x2 rw[3][3];
#pragma unroll
        for (int rr = 0; rr < 3; ++rr)
#pragma unroll
          for (int dx = 0; dx < 3; ++dx)
            rw[rr][dx] = f32x2{rp[(rr * T::RW + dx) * T::S1], rp[(rr * T::RW + dx) * T::S1 + 16]};
        rp += 3 * T::RW * T::S1;
        float* gdst = Gs + sx * T::SG + cp;
        auto step = [&](f32x2 (&p0)[3], f32x2 (&p1)[3], f32x2 (&p2)[3], const int y) {
          const float keep = PL ? 1.f : ((colok && y < nrows) ? 1.f : 0.f);
          const f32x2 xraw = RC ? f32x2{gdst[0], gdst[16]} : xpv[y];       // RC: parked where this thread's dP goes a few lines down
          const f32x2 xc = PL ? xraw : xraw * (xm * keep);
          f32x2 t = wf[0] * p0[0];
          t = __builtin_elementwise_fma(wf[1], p0[1], t);
          t = __builtin_elementwise_fma(wf[2], p0[2], t);
#pragma unroll
          for (int dx = 0; dx < 3; ++dx) dwa[dx] = __builtin_elementwise_fma(xc, p0[dx], dwa[dx]);
          // p0 is free now: fetch ring row y + 3 (the last fetches read past the ring: inside smem, never used)
#pragma unroll
          for (int dx = 0; dx < 3; ++dx) p0[dx] = f32x2{rp[dx * T::S1], rp[dx * T::S1 + 16]};
          rp += T::RW * T::S1;
#pragma unroll
          for (int kx = 0; kx < 3; ++kx) {
            t = __builtin_elementwise_fma(wf[3 + kx], p1[kx], t);
            t = __builtin_elementwise_fma(wf[6 + kx], p2[kx], t);
            dwa[3 + kx] = __builtin_elementwise_fma(xc, p1[kx], dwa[3 + kx]);
            dwa[6 + kx] = __builtin_elementwise_fma(xc, p2[kx], dwa[6 + kx]);
          }
          gdst[0] = PL ? t.x : t.x * keep;                  // pixels outside the image contribute nothing to dW1 / dT
          gdst[16] = PL ? t.y : t.y * keep;
          gdst += T::TW * T::SG;
        };
        step(rw[0], rw[1], rw[2], 0);
        step(rw[1], rw[2], rw[0], 1);
        step(rw[2], rw[0], rw[1], 2);
        step(rw[0], rw[1], rw[2], 3);
        step(rw[1], rw[2], rw[0], 4);
        step(rw[2], rw[0], rw[1], 5);
        step(rw[0], rw[1], rw[2], 6);
        step(rw[1], rw[2], rw[0], 7);
      };
      if (plain) walk(std::true_type{});
      else walk(std::false_type{});
      UIR_ST(3);
      if constexpr (!RC) load_xp(cn, cp, sx, last ? tn : tc);
      UIR_ST(9);
      // sum over the wave's four strips = the four lanes of a quad: two DPP adds per value (quad_perm [1,0,3,2], [2,3,0,1]); the strips
      // on lanes 16 apart needed a v_permlane16_swap + v_permlane32_swap pair with its register copies (~110 VALU per chunk instead
      // of ~40).  Then one row per wave in LDS
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        const float vx = dpp_add<0x4E>(dpp_add<0xB1>(dwa[t].x)), vy = dpp_add<0x4E>(dpp_add<0xB1>(dwa[t].y));
        if ((lane & 3) == 0) {
          WR[(wave * 16 + cp) * 18 + t] = vx;
          WR[(wave * 16 + cp) * 18 + 9 + t] = vy;
        }
      }
      UIR_ST(10);
      lds_barrier2();
      UIR_ST(4);
      {
        const int e2 = t_ < 32 ? t_ + 256 : t_;             // (threads >= 32 read a valid slot again; their second sum is unused)
        const float va = (WR[t_] + WR[288 + t_]) + (WR[576 + t_] + WR[864 + t_]);
        const float vb = (WR[e2] + WR[288 + e2]) + (WR[576 + e2] + WR[864 + e2]);
        DA_ADD(ci, va, vb);
      }
      UIR_ST(5);

      // ---- (c) dT += dP chunk . W1 chunk (row tile = wave) ----
      {
        const float* gr = Gs + (wave * 32 + li) * T::SG + 4 * h2;
        if constexpr (RC) {
          // split-bf16 (see the XP tile above): k slot t of step s2 is channel 16 s2 + 8 (t / 4) + 4 h2 + t % 4 = w1v[8 s2 + t]
#pragma unroll
          for (int s2 = 0; s2 < 2; ++s2) {
            const f32x4v a0 = *reinterpret_cast<const f32x4v*>(gr + 16 * s2), a1 = *reinterpret_cast<const f32x4v*>(gr + 16 * s2 + 8);
            const float wb[8] = {w1v[8 * s2], w1v[8 * s2 + 1], w1v[8 * s2 + 2], w1v[8 * s2 + 3], w1v[8 * s2 + 4], w1v[8 * s2 + 5], w1v[8 * s2 + 6], w1v[8 * s2 + 7]};
            acc = uir_mfma_split(uir_split8(make_float4(a0[0], a0[1], a0[2], a0[3]), make_float4(a1[0], a1[1], a1[2], a1[3])), uir_split8(wb), acc);
          }
        } else {
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
          const f32x4v v = *reinterpret_cast<const f32x4v*>(gr + 8 * jj);
#pragma unroll
          for (int e = 0; e < 4; ++e) acc = uir_mfma(v[e], w1v[4 * jj + e], acc);
        }
        }
      }
      UIR_ST(6);
      // ---- (d) dW1 chunk granule += dP^T . (xhat gamma + beta) over this wave's pixels.  k-slot j of step s is pixel
      //      16 (s >> 2) + 4 j + (s & 3): a lane's four steps of a group read four CONSECUTIVE pixels - one ds_read_b128 of the transposed
      //      xhat for the B operand; the two A rows a 32-lane half reads sit 4 * 36 floats = 16 banks apart ----
      {
        const int p0 = 4 * j16 + 64 * part * (T::STEPS / 16);
        const float* ga = Gs + 16 * gr_k + l16 + p0 * T::SG;
        const float* xb = XH + n16 * T::SXT + p0;
        f32x4v d, d1 = f32x4v{0.f, 0.f, 0.f, 0.f};             // two chains: a dependent v_mfma_f32_16x16x4_f32 issues every 40 cycles, not 32
        DW_GET(ci, d);
        constexpr int NGR = T::STEPS / 8;                      // groups of 8 steps = 32 pixels; operands of group g + 1 are read
        float av[2][8];                                        // from LDS before the MFMAs of group g are issued
        f32x4v bq[2][2];
        auto fetch = [&](int g, float (&ao)[8], f32x4v (&bo)[2]) {
#pragma unroll
          for (int s = 0; s < 8; ++s) ao[s] = ga[(32 * g + 16 * (s >> 2) + (s & 3)) * T::SG];   // compile-time LDS offsets
          bo[0] = *reinterpret_cast<const f32x4v*>(xb + 32 * g);
          bo[1] = *reinterpret_cast<const f32x4v*>(xb + 32 * g + 16);
        };
        // sched_barriers: as plain code hipcc sank every ds_read next to the MFMA that consumes it - one exposed LDS round trip per
        // MFMA pair, 26 % of the kernel (tools/probes/stamps, tools/scan_isa_chains.py); pinned, a group's reads land under the
        // previous group's 8 MFMAs
        fetch(0, av[0], bq[0]);
#pragma unroll
        for (int g = 0; g < NGR; ++g) {
          if (g + 1 < NGR) fetch(g + 1, av[(g + 1) & 1], bq[(g + 1) & 1]);
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int s = 0; s < 8; s += 2) {
            d = mfma16(av[g & 1][s], fmaf(bq[g & 1][s >> 2][s & 3], gam16, bet16), d);
            d1 = mfma16(av[g & 1][s + 1], fmaf(bq[g & 1][(s + 1) >> 2][(s + 1) & 3], gam16, bet16), d1);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
        d += d1;
        DW_PUT(ci, d);
      }
      UIR_ST(7);
    }

    // ---- tile epilogue: LayerNorm backward on the dT accumulators (lane = channel, registers = pixels of row tile `wave`).
    //      Register r is tile row 2 wave + (r >> 3), column (r & 3) + 8 ((r >> 2) & 1) + 4 h2.  All 16 residual-gradient loads are
    //      issued first ----
    {
      UIR_DWLN_COORDS;
      const int nn = li < C ? li : 0;
      const float invc = 1.0f / (float)C;
      unsigned mo[16];
      float rv[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int iy = y0 + 2 * wave + (r >> 3), ix = x0 + (r & 3) + 8 * ((r >> 2) & 1) + 4 * h2;
        const unsigned m = img + (unsigned)((iy < a.H ? iy : a.H - 1) * a.Wd + (ix < a.Wd ? ix : a.Wd - 1));
        mo[r] = (iy < a.H && ix < a.Wd && li < C) ? m : ~0u;
        rv[r] = ldg_off(a.dRes, (m * (unsigned)a.ldr + nn) * 4u);
      }
      // Passes - LDS operands, arithmetic (two halves of 8 rows: registers), stores.  As one loop every store sat under its own
      // branch, and since vmcnt retires in issue order and the compiler cannot count a store it may have skipped, each iteration's
      // wait for rv[r] was vmcnt(0): it also waited for the PREVIOUS iteration's store to be acknowledged - 16 HBM round trips per
      // tile, 11 % of the kernel.  The asm pins keep LLVM from sinking the arithmetic (and with it the rv waits) back into the
      // store branches.
#pragma unroll
      for (int hf = 0; hf < 2; ++hf) {
        float xqv[8], rsv[8];
#pragma unroll
        for (int q4 = 0; q4 < 2; ++q4) {                     // registers 8 hf + 4 q4 .. + 3: four consecutive pixels of tile row 2 wave + hf
          const int p = (2 * wave + hf) * T::TW + 8 * q4 + 4 * h2;
          const f32x4v xv4 = *reinterpret_cast<const f32x4v*>(XH + nn * T::SXT + p);
          const f32x4v rs4 = *reinterpret_cast<const f32x4v*>(RS + p);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            xqv[4 * q4 + e] = xv4[e];
            rsv[4 * q4 + e] = rs4[e];
          }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          const int r = 8 * hf + q;
          const float xq = li < C ? xqv[q] : 0.f;
          const float dt = acc[r];
          const float g = dt * gam_n;
          const float s1 = swap16_sum(row_sum16(g)), s2 = swap16_sum(row_sum16(g * xq));   // over the 32 channel lanes of the half
          float dx = rsv[q] * (g - s1 * invc - xq * (s2 * invc)) + rv[r];
          asm volatile("" : "+v"(dx));
          acc[r] = dx;
          dgam = fmaf(dt, xq, dgam);
          dbet += dt;
        }
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int r = 0; r < 16; ++r)
        if (mo[r] != ~0u) a.dX[mo[r] * (unsigned)a.ldo + nn] = acc[r];
    }
    UIR_ST(8);
  }
  UIR_ST_FLUSH();

  // ---- slab row of this workgroup: dWdw (CHreal x 9) | dW1 (CHreal x C) | dgamma (C) | dbeta (C) ----
  UIR_DWLN_COORDS;
  const int chreal = SPLIT ? 2 * h : ch;
  float* slab = a.slab + (long)blockIdx.x * a.slab_stride;
  float* sw1 = slab + (long)chreal * 9;
  lds_barrier2();                                            // every wave is past its last reads of GS1 / Gs / XH
  if (T::PARTS == 2) {                                       // C = 16: the two pixel parts of a granule meet in LDS
    if (part == 1) {
#pragma unroll 1
      for (int ci = 0; ci < NCH; ++ci) {
        f32x4v d;
        DW_GET(ci, d);
#pragma unroll
        for (int r = 0; r < 4; ++r) GS1[((ci * 2 + gr_k) * 4 + r) * 64 + lane] = d[r];
      }
    }
    lds_barrier2();
  }
  if (part == 0) {
#pragma unroll 1
    for (int ci = 0; ci < NCH; ++ci) {
      f32x4v d;
      DW_GET(ci, d);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float v = d[r];
        if (T::PARTS == 2) v += GS1[((ci * 2 + gr_k) * 4 + r) * 64 + lane];
        const int k = 16 * gr_k + 4 * j16 + r;               // chunk-local channel: row of the 16 x 16 result
        const bool ok = n16 < C && (k < 16 ? okA(ci, k) : okB(ci, k - 16));
        const long row = k < 16 ? rowA(ci, k) : rowB(ci, k - 16);
        if (ok) sw1[row * C + n16] = v;
      }
    }
  }
#undef DW_GET
#undef DW_PUT
#pragma unroll 1
  for (int ci = 0; ci < NCH; ++ci) {
    float va, vb;
    DA_GET(ci, va, vb);
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      const int rem = t_ + 256 * half;
      if (rem < 288) {
        const int c = rem / 18, q = rem - c * 18;
        const int grp = q / 9, t = q - grp * 9;
        const bool ok = grp == 0 ? okA(ci, c) : okB(ci, c);
        const long row = grp == 0 ? rowA(ci, c) : rowB(ci, c);
        if (ok) slab[row * 9 + (8 - t)] = half == 0 ? va : vb;
      }
    }
  }
#undef DA_ADD
#undef DA_GET
  {
    const float g = dgam + __shfl_xor(dgam, 32), bsum = dbet + __shfl_xor(dbet, 32);
    lds_barrier2();
    if (h2 == 0) {
      WR[(wave * 2 + 0) * 32 + li] = g;
      WR[(wave * 2 + 1) * 32 + li] = bsum;
    }
    lds_barrier2();
    float* sgb = sw1 + (long)chreal * C;
    for (int e = t_; e < 2 * C; e += 256) {
      const int which = e / C, n = e - which * C;
      float s = 0.f;
      for (int w = 0; w < 4; ++w) s += WR[(w * 2 + which) * 32 + n];
      sgb[which * C + n] = s;
    }
  }
}
#undef UIR_DWLN_COORDS

template <int C, int NCH, bool SPLIT>
int dwln_launch(const UirDwLnBwdArgs& a, hipStream_t st) {
  using T = DwLn<C, NCH, SPLIT>;
  if (a.XP) hipLaunchKernelGGL((uir_dwln_bwd_kernel<C, NCH, SPLIT, false>), dim3(a.nblocks), dim3(256), (size_t)T::LDS_FLOATS * 4, st, a);
  else hipLaunchKernelGGL((uir_dwln_bwd_kernel<C, NCH, SPLIT, true>), dim3(a.nblocks), dim3(256), (size_t)T::LDS_FLOATS * 4, st, a);
  return uir_launch_status();
}

}  // namespace
UIR_ST_EXPORT

extern "C" int uir_dwln_bwd(const UirDwLnBwdArgs* p, void* stream) {
  const UirDwLnBwdArgs& a = *p;
  UIR_REQUIRE(a.XP || uir_fused_split() != 0);      // the recompute of XP runs on split-bf16 MFMA (UIR_FUSED_SPLIT=0: pass the stored tensor)
  UIR_REQUIRE(a.G && a.Wdw && a.W1 && a.X && a.stats && a.gamma && a.beta && a.dRes && a.dX && a.slab);
  UIR_REQUIRE(a.B > 0 && a.H > 0 && a.Wd > 0 && a.nblocks > 0 && (a.C == 16 || a.C == 32));
  UIR_REQUIRE((a.ldg & 3) == 0 && (a.ldx & 3) == 0 && uir_aligned16(a.G) && uir_aligned16(a.X) && (a.ldx >= a.C) && a.ldr >= a.C && a.ldo >= a.C);
  UIR_REQUIRE((reinterpret_cast<uintptr_t>(a.stats) & 7u) == 0);
  const bool split = a.hp > 0;
  UIR_REQUIRE(a.il == 0 || (a.il == 1 && split));
  const int chreal = split ? 2 * a.h : a.ch;
  if (split) UIR_REQUIRE(a.h > 0 && a.hp >= a.h && (a.hp & 15) == 0 && a.ldg >= 2 * a.hp && (!a.XP || a.ldp >= 2 * a.hp));
  else UIR_REQUIRE(a.ch > 0 && (a.ch & 15) == 0 && a.ldg >= a.ch && (!a.XP || a.ldp >= a.ch));
  UIR_REQUIRE(a.slab_stride >= (long)chreal * 9 + (long)chreal * a.C + 2 * a.C);
  // 32-bit byte offsets into G
  {
    const long rows = (long)a.B * a.H * a.Wd;
    const long ldmax = (a.XP && a.ldp > a.ldg) ? a.ldp : a.ldg;
    UIR_REQUIRE(rows * ldmax < (1L << 30) && rows * a.ldx < (1L << 30) && rows * a.ldr < (1L << 30) && rows * a.ldo < (1L << 30));
  }
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int nch = split ? a.hp / 16 : (a.ch + 31) / 32;
#define UIR_DWLN(CV, NV, SV) return dwln_launch<CV, NV, SV>(a, st)
  if (split) {
    if (a.C == 32 && nch == 6) UIR_DWLN(32, 6, true);
    if (a.C == 16 && nch == 3) UIR_DWLN(16, 3, true);
  } else {
    if (a.C == 32 && nch == 3) UIR_DWLN(32, 3, false);
    if (a.C == 16 && nch == 2) UIR_DWLN(16, 2, false);
  }
#undef UIR_DWLN
  return UIR_EINVAL;
}
