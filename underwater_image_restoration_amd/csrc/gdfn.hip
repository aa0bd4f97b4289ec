// uir_gdfn_fwd: the whole GDFN forward of one TransformerBlock in ONE pass over the pixels
//   y = x + W_out . ( gelu(dw3x3(W_in . LN(x))[x1 half]) * dw3x3(W_in . LN(x))[x2 half] )
// (SpectralTransformer.py:127-130 with the norm2 / residual of :147-148).
//
// One workgroup owns a TH x TW pixel tile of one image.  The block input is read once (tile + a
// one-pixel ring), LayerNorm'd in registers and kept there in MFMA A-operand layout (pixel on the
// accumulator row, channel on the contraction index).  The hidden dimension is walked in chunks of
// 16 x1-channels + their 16 x2 partners (one 32-column MFMA tile):
//   (a) t1 chunk = LN(x) . W_in^T            f32 MFMA, K = C, on tile + ring        -> LDS
//   (b) t2 = depthwise 3x3 of t1 from LDS (packed-fp32 VALU)                        -> LDS
//   (c) y += (gelu(t2a) * t2b) . W_out^T     gate applied on the A-operand load; f32 MFMA, K = 16,
//                                            accumulators live across chunks
// so the two 2*hp-column intermediates never round-trip through HBM between the three ops.  For
// training they are still WRITTEN once (the backward kernels read them), as 16-byte stores staged
// through the LDS copies (a dword store per value costs ~75 SIMD cycles per wave instruction:
// measured +260 us per 1.6 GB); inference passes T1 = T2 = NULL and the kernel moves 2*C floats
// per pixel instead of ~27*C.
// f32 MFMA and VALU work of two waves on one SIMD do NOT overlap on gfx950
// (tools/probes/mfma_valu_overlap.hip: MFMA-only wave + VALU-only wave = sum of both), so the
// kernel's floor is MFMA time + VALU time, not their maximum.
// Out-of-image ring pixels get LN(x) := 0, hence t1 = 0 there: exactly Conv2d's zero padding
// (project_in has no bias).
#include "uir_common.h"
#include "../../include/uir_kernels.h"
#include "uir_stamps.h"

namespace {

template <int C, int TH, int TW>
struct GdfnTile {
  static constexpr int RW = TW + 2, RH = TH + 2;
  static constexpr int NPX1 = RW * RH;              // tile + ring pixels
  static constexpr int NRT1 = (NPX1 + 31) / 32;     // 32-pixel MFMA row tiles over tile + ring
  static constexpr int RTW = (NRT1 + 3) / 4;        // ... per wave
  static constexpr int NI = TH * TW;                // interior pixels (multiple of 32)
  static constexpr int NRTI = NI / 32;
  static constexpr int UNITS = NRTI * ((C + 31) / 32);  // (c) work units
  static constexpr int UPW = (UNITS + 3) / 4;
  static constexpr int S1 = 40;                     // LDS row stride of the t1 chunk (32 cols + 8)
  static constexpr int SG = 36;                     // LDS row stride of the t2 chunk (16 + 16 cols + 4)
  static constexpr int NT = (C + 31) / 32;          // 32-column output tiles
  static constexpr int KJ = C / 8;                  // float4 groups along K=C per lane half
  static constexpr int SEG = 16 / TW;               // vertical segments of the depthwise walk
  static constexpr int ROWS = TH / SEG;
  static constexpr int LDS_FLOATS = NRT1 * 32 * S1 + NI * SG;
  static constexpr int NST = (NI * 8 + 255) / 256;  // float4 write-out passes (8 lanes per pixel row of 32 floats)
  static_assert(NI % 32 == 0 && (16 % TW) == 0 && TH % SEG == 0 && C % 8 == 0 && 4 % NT == 0, "tile shape");
};

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains vmcnt, i.e. every
// in-flight global store (the t1/t2 write-out) and load (the next chunk's weights) would be waited
// for at each of the two barriers per chunk.
// t1/t2 are next read by the backward pass, a whole forward later: streaming (nt) stores keep them from evicting the
// tile inputs and weights in L2 (A/B at C=16, 256x256: 309 vs 376 us; C=32: 661 vs 684 us)
__device__ __forceinline__ void st4(char* p, const float4& v) {
  __builtin_nontemporal_store(f32x4{v.x, v.y, v.z, v.w}, reinterpret_cast<f32x4*>(p));
}
__device__ __forceinline__ void pin(float& x) { asm volatile("" : "+v"(x)); }   // value must be in its VGPR here
// `ok ? load : 0` written as a select lets hipcc put the load under an exec branch (a conditional VMEM instruction and its wait);
// loaded from a clamped address and masked by an AND the load is unconditional
__device__ __forceinline__ float4 keep4(const float4& t, bool ok) {
  const unsigned m = ok ? 0xffffffffu : 0u;
  return make_float4(__uint_as_float(__float_as_uint(t.x) & m), __uint_as_float(__float_as_uint(t.y) & m),
                     __uint_as_float(__float_as_uint(t.z) & m), __uint_as_float(__float_as_uint(t.w) & m));
}
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));      // four consecutive floats at a 4-byte aligned address
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }


template <int C, int TH, int TW, bool STORE, bool SB>
__global__ __launch_bounds__(256, 2) void uir_gdfn_fwd_kernel(const UirGdfnFwdArgs a) {
  using T = GdfnTile<C, TH, TW>;
  extern __shared__ float smem[];
  float* T1s = smem;
  float* Gs = smem + T::NRT1 * 32 * T::S1;
  float* Ws = smem + T::LDS_FLOATS;       // wide levels: two staged W_in chunks [32 rows][C + 4]
  // split-bf16 form: ALL depthwise taps of the launch staged once per workgroup, row of hidden channel cc = 9 taps of x1's channel |
  // 9 of x2's; they were 18 dword loads per lane and chunk (tools/probes/stamps: the weight loads' issue stalls behind the write-out
  // stores were 11.6 % of the kernel at C = 32; at C >= 64, with one or two workgroups per CU, each chunk waited out their latency).
  // Row stride 20 floats (16-byte rows, b128 reads) where LDS allows it (C <= 32), 18 (b64 reads) beside the staged W_in chunks
  // C = 64 keeps them too (row stride 18, b64 reads, behind the two staged f32 W_in chunks): no faster, but the kernel spills without;
  // C = 128 has no LDS for them beside its bf16 weight planes
  constexpr bool TAPS_LDS = SB && C <= 64;
  constexpr int SWT = C <= 32 ? 20 : 18;
  float* WT = smem + T::LDS_FLOATS + (C > 32 ? 2 * 32 * (C + 4) : 0);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, h2 = lane >> 5;
  const int tiles_x = (a.Wd + TW - 1) / TW, tiles_y = (a.H + TH - 1) / TH;
  int bid = blockIdx.x;
  const int tx = bid % tiles_x;
  bid /= tiles_x;
  const int ty = bid % tiles_y;
  const int b = bid / tiles_y;
  const int y0 = ty * TH, x0 = tx * TW;
  const long img = (long)b * a.H * a.Wd;
  const int h = a.h, hp = a.hp;

  UIR_ST_DECL;
  // hp * 18 tap slots: <= 6.75 per thread at hp <= 96 (C <= 32: registers across phase 0); C >= 64 (hp <= 352: 24.75 per thread)
  // copies them in groups of four before phase 0
  constexpr int NTAP = !TAPS_LDS ? 1 : C <= 32 ? 7 : 4;
  float tapv[NTAP];
  auto tap_load = [&](int it0) {
#pragma unroll
    for (int it = 0; it < NTAP; ++it) {
      const int e = tid + 256 * (it0 + it), cc = e / 18, q = e - 18 * cc;
      const bool ok = cc < h;
      tapv[it] = a.Wdw[ok ? (q < 9 ? cc * 9 + q : (h + cc) * 9 + (q - 9)) : 0];
    }
  };
  auto tap_store = [&](int it0) {
#pragma unroll
    for (int it = 0; it < NTAP; ++it) {
      const int e = tid + 256 * (it0 + it), cc = e / 18, q = e - 18 * cc;
      if (cc < hp) WT[cc * SWT + q] = cc < h ? tapv[it] : 0.f;
    }
  };
  if constexpr (TAPS_LDS) {
    if constexpr (C <= 32) {
      tap_load(0);
    } else {
      for (int it0 = 0; it0 * 256 < hp * 18; it0 += NTAP) {
        tap_load(it0);
        tap_store(it0);
      }
    }
  }
  // ---- phase 0: x on tile + ring -> LayerNorm -> A-operand registers (k = 8jj + 4h2 + e) ----
  float A[T::RTW][C / 2];
  {
    float4 gam[T::KJ], bet[T::KJ];
#pragma unroll
    for (int jj = 0; jj < T::KJ; ++jj) {
      gam[jj] = *reinterpret_cast<const float4*>(a.ln_g + 8 * jj + 4 * h2);
      bet[jj] = *reinterpret_cast<const float4*>(a.ln_b + 8 * jj + 4 * h2);
    }
#pragma unroll
    for (int q = 0; q < T::RTW; ++q) {
      const int r = wave + 4 * q;
      if (r < T::NRT1) {
        const int p = r * 32 + li;
        const int ry = p / T::RW, rx = p - ry * T::RW;
        const int iy = y0 + ry - 1, ix = x0 + rx - 1;
        const bool valid = p < T::NPX1 && iy >= 0 && iy < a.H && ix >= 0 && ix < a.Wd;
        const long m = valid ? img + (long)iy * a.Wd + ix : img;
        const float* xr = a.X + m * a.ldx + 4 * h2;
        float4 v[T::KJ];
        float s = 0.f;
#pragma unroll
        for (int jj = 0; jj < T::KJ; ++jj) v[jj] = *reinterpret_cast<const float4*>(xr + 8 * jj);
        __builtin_amdgcn_sched_barrier(0);          // all of the row's loads in flight before the first is summed (one variant waited for each)
#pragma unroll
        for (int jj = 0; jj < T::KJ; ++jj) s += (v[jj].x + v[jj].y) + (v[jj].z + v[jj].w);
        s += __shfl_xor(s, 32);
        const float mean = s / (float)C;
        float qq = 0.f;
#pragma unroll
        for (int jj = 0; jj < T::KJ; ++jj) {
          const float d0 = v[jj].x - mean, d1 = v[jj].y - mean, d2 = v[jj].z - mean, d3 = v[jj].w - mean;
          qq += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
        }
        qq += __shfl_xor(qq, 32);
        const float rstd = rsqrtf(qq / (float)C + a.ln_eps);
        const float keep = valid ? 1.f : 0.f;
#pragma unroll
        for (int jj = 0; jj < T::KJ; ++jj) {
          A[q][4 * jj + 0] = keep * fmaf((v[jj].x - mean) * rstd, gam[jj].x, bet[jj].x);
          A[q][4 * jj + 1] = keep * fmaf((v[jj].y - mean) * rstd, gam[jj].y, bet[jj].y);
          A[q][4 * jj + 2] = keep * fmaf((v[jj].z - mean) * rstd, gam[jj].z, bet[jj].z);
          A[q][4 * jj + 3] = keep * fmaf((v[jj].w - mean) * rstd, gam[jj].w, bet[jj].w);
        }
        const bool interior = ry >= 1 && ry <= TH && rx >= 1 && rx <= TW;
        if (a.stats && h2 == 0 && valid && interior)
          *reinterpret_cast<float2*>(a.stats + m * 2) = make_float2(mean, rstd);
      }
    }
  }

  if constexpr (TAPS_LDS && C <= 32) tap_store(0);   // visible after chunk 0's first barrier, read first in its (b)
  // SB: the A operand as three bf16 planes per 16-deep k step, split ONCE per tile (it is the same for every chunk)
  UirBf3 Ap[SB ? T::RTW : 1][SB ? C / 16 : 1];
  if constexpr (SB) {
#pragma unroll
    for (int q = 0; q < T::RTW; ++q)
#pragma unroll
      for (int s = 0; s < C / 16; ++s) {
        const float x[8] = {A[q][8 * s], A[q][8 * s + 1], A[q][8 * s + 2], A[q][8 * s + 3], A[q][8 * s + 4], A[q][8 * s + 5], A[q][8 * s + 6], A[q][8 * s + 7]};
        Ap[q][s] = uir_split8(x);
      }
  }

  // (c) work units (interior row tile, 32-column output tile): unit u = wave + 4 j -> rt = u / NT, nt = u % NT;
  // NT divides 4, so a wave keeps ONE output tile nt = wave % NT and its W_out columns
  f32x16 acc[T::UPW];
#pragma unroll
  for (int j = 0; j < T::UPW; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  const int my_n = (wave % T::NT) * 32 + li;

  // depthwise-walk coordinates of this thread: channel pair cp, column sx, rows [ys, ys + ROWS)
  // The four 16-lane groups of a wave take columns {0, 2, 1, 3} of the wave's four: the two groups of a
  // 32-lane half then sit 2 * S1 = 80 floats apart = 16 banks (ds_read_b32 banks are address % 32), conflict-free.
  const int cp = tid & 15;
  const int strip = (tid >> 6) * 4 + (((lane >> 4) & 1) << 1 | (lane >> 5));
  const int sx = strip % TW, ys = (strip / TW) * T::ROWS;

  auto load_w = [&](int c0, float4 (&win)[T::KJ], float (&wo)[8], float (&w1)[9], float (&w2)[9]) {
    // clamped loads, zeroed by select: no divergent branches
    {
      const int cc = c0 + (li & 15);
      const bool ok = cc < h;
      const long row = ok ? (li < 16 ? cc : h + cc) : 0;
      const float* wr = a.Win + row * C + 4 * h2;
#pragma unroll
      for (int jj = 0; jj < T::KJ; ++jj) {
        const float4 t = *reinterpret_cast<const float4*>(wr + 8 * jj);
        win[jj] = ok ? t : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      const int k = c0 + 8 * (s >> 2) + 4 * h2 + (s & 3);
      const bool ok = my_n < C && k < h;
      const float t = a.Wout[ok ? (long)my_n * h + k : 0];
      wo[s] = ok ? t : 0.f;
    }
    {
      const int cc = c0 + cp;
      const bool ok = cc < h;
      const float* p1 = a.Wdw + (long)(ok ? cc : 0) * 9;
      const float* p2 = a.Wdw + (long)(ok ? h + cc : 0) * 9;
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        const float u1 = p1[t], u2 = p2[t];
        w1[t] = ok ? u1 : 0.f;
        w2[t] = ok ? u2 : 0.f;
      }
    }
  };
  // write-out slots of this thread: float4 j of pixel p (8 lanes cover the 16 x1 + 16 x2 floats of a pixel);
  // byte offset into T1/T2 without the chunk term (fits 32 bits: checked by the launcher), ~0 = not written
  unsigned st_off[STORE ? T::NST : 1];
  int st_lds[STORE ? T::NST : 1];
  if (STORE) {
#pragma unroll
    for (int it = 0; it < T::NST; ++it) {
      const int e = tid + 256 * it, p = e >> 3, j = e & 7;
      const int py = p / TW, px = p - py * TW;
      const int iy = y0 + py, ix = x0 + px;
      const bool ok = p < T::NI && iy < a.H && ix < a.Wd;
      // a.il: the chunk's 16 x1 + 16 x2 values are 32 CONSECUTIVE columns (one 128-byte line per pixel and chunk)
      st_off[it] = ok ? (unsigned)(((img + (long)iy * a.Wd + ix) * a.ldt + (a.il ? 4 * j : (j < 4 ? 4 * j : hp + 4 * (j - 4)))) * 4) : ~0u;
      st_lds[it] = ((py + 1) * T::RW + px + 1) * T::S1 + 4 * j;
    }
  }

  // C <= 32: the next chunk's weights are prefetched into a second register set.  Wider levels have no registers for
  // that (W_in alone is C/2 per lane): their W_in chunk goes through LDS, double-buffered - fetched one chunk ahead
  // with the other small operands loaded one phase ahead into the registers their last use just released.
  constexpr bool PF = C <= 32;
  constexpr int NPF = PF ? T::KJ : 1;
  constexpr int WS = C + 4;                       // staged row stride (floats): 16-byte aligned, conflict-free b128 reads
  constexpr int NSTG = PF ? 1 : (32 * (C / 4)) / 256;   // float4 per thread and chunk
  float4 win[PF ? T::KJ : 1], win_n[NPF];
  float wo[8], wo_n[8], w1[9], w2[9], w1_n[9], w2_n[9];
  float4 stg[NSTG];
  auto stage_load = [&](int c0) {                 // global -> registers (rows of the chunk: 16 x1 rows | 16 x2 rows)
#pragma unroll
    for (int it = 0; it < NSTG; ++it) {
      const int e = tid + 256 * it, r = e / (C / 4), q = e - r * (C / 4);
      const int cc = c0 + (r & 15);
      const bool ok = cc < h;
      stg[it] = keep4(*reinterpret_cast<const float4*>(a.Win + (long)(ok ? (r < 16 ? cc : h + cc) : 0) * C + 4 * q), ok);
    }
  };
  // SB, wide levels: the staging threads split the chunk into its three bf16 planes ONCE (each of the four waves splitting the rows it
  // reads was 640 VALU instructions per wave and chunk at C = 128, ~45 % of a chunk).  Plane p of buffer b: [32 rows][C + 8] bf16, a
  // lane's eight values of k step s - k = 16 s + 4 h2 + e and 16 s + 8 + 4 h2 + e, the order the A planes use - contiguous: the
  // float4 at k = 16 s + 8 g + 4 h goes to slot 16 s + 8 h + 4 g
  constexpr bool WPRE = SB && C >= 128;              // (C = 64: four k steps per chunk - the split is a small part there, and LDS is full)
  constexpr int RSB = C + 8;                       // plane row stride (bf16 units): 16-byte rows, (C + 8) / 2 = 4 mod 32 dwords
  constexpr int PLSZ = 32 * RSB;                   // one plane (bf16 units)
  __bf16* Wp = reinterpret_cast<__bf16*>(Ws);
  auto stage_store = [&](int buf) {               // registers -> LDS
#pragma unroll
    for (int it = 0; it < NSTG; ++it) {
      const int e = tid + 256 * it, r = e / (C / 4), q = e - r * (C / 4);
      if constexpr (WPRE) {
        const UirBf3x4 pl = uir_split4(stg[it]);
        const int slot = 16 * (q >> 2) + 8 * (q & 1) + 4 * ((q >> 1) & 1);
        __bf16* d = Wp + buf * 3 * PLSZ + r * RSB + slot;
        *reinterpret_cast<uint2*>(d) = pl.h;
        *reinterpret_cast<uint2*>(d + PLSZ) = pl.m;
        *reinterpret_cast<uint2*>(d + 2 * PLSZ) = pl.l;
      } else {
        *reinterpret_cast<float4*>(Ws + buf * 32 * WS + r * WS + 4 * q) = stg[it];
      }
    }
  };
  auto load_small = [&](int c0, float (&wo_)[8], float (&w1_)[9], float (&w2_)[9]) {
    // W_out[my_n][k] for the lane's two groups of four consecutive k: ONE 16-byte load per group (4-byte aligned: a row is h floats)
    // wherever the whole group is real - every group but the last chunk's; 8 dword loads with 64-bit address arithmetic cost the
    // chunk ~10 % of its time in issue stalls behind the write-out stores (tools/probes/stamps)
    const unsigned wrow = (unsigned)(my_n < C ? my_n : 0) * (unsigned)h;
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      const int k0 = c0 + 8 * g + 4 * h2;
      if (k0 + 4 <= h) {
        const f32x4u t = *reinterpret_cast<const f32x4u*>(a.Wout + (wrow + (unsigned)k0));
        const unsigned m = my_n < C ? 0xffffffffu : 0u;
#pragma unroll
        for (int e = 0; e < 4; ++e) wo_[4 * g + e] = __uint_as_float(__float_as_uint(t[e]) & m);
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const bool ok = my_n < C && k0 + e < h;
          const float t = a.Wout[ok ? wrow + (unsigned)(k0 + e) : 0u];
          wo_[4 * g + e] = ok ? t : 0.f;
        }
      }
    }
    if constexpr (!TAPS_LDS) {
      const int cc = c0 + cp;
      const bool ok = cc < h;
      const float* p1 = a.Wdw + (long)(ok ? cc : 0) * 9;
      const float* p2 = a.Wdw + (long)(ok ? h + cc : 0) * 9;
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        const float u1 = p1[t], u2 = p2[t];
        w1_[t] = ok ? u1 : 0.f;
        w2_[t] = ok ? u2 : 0.f;
      }
    }
  };
  auto load_win = [&](int c0, float4 (&win_)[NPF]) {   // the W_in fragment alone (SB, C <= 32)
    const int cc = c0 + (li & 15);
    const bool ok = cc < h;
    const float* wr = a.Win + (long)(ok ? (li < 16 ? cc : h + cc) : 0) * C + 4 * h2;
#pragma unroll
    for (int jj = 0; jj < NPF; ++jj) win_[jj] = keep4(*reinterpret_cast<const float4*>(wr + 8 * jj), ok);
  };
  // SB: only W_in is fetched a chunk ahead (it is needed first, at the top of (a)); W_out and the depthwise taps of the CURRENT chunk
  // are loaded at its top and land under (a) - the bf16 planes of the A operand take the registers of the second small-weight set
  if constexpr (SB) {
    if constexpr (PF) {
      load_win(0, win_n);
#pragma unroll
      for (int jj = 0; jj < NPF; ++jj) win[jj] = win_n[jj];
    } else {
      stage_load(0);
      load_small(0, wo, w1, w2);
      stage_store(0);
      lds_barrier();
    }
  } else if constexpr (PF) {
    load_w(0, win, wo, w1, w2);
  } else {
    stage_load(0);
    load_small(0, wo, w1, w2);
    stage_store(0);
    lds_barrier();
  }

  int wbuf = 0;
  UIR_ST(0);
  for (int c0 = 0; c0 < hp; c0 += 16) {
    if constexpr (SB) {
      if constexpr (PF) {
        load_win(c0 + 16 < hp ? c0 + 16 : c0, win_n);
        load_small(c0, wo, w1, w2);
      } else {
        // wide levels (one or two workgroups per CU): nothing hides a load issued and consumed inside one chunk - W_out's fragment
        // is fetched a chunk ahead too (the taps live in LDS)
        stage_load(c0 + 16 < hp ? c0 + 16 : c0);
        load_small(c0 + 16 < hp ? c0 + 16 : c0, wo_n, w1_n, w2_n);
      }
    } else if constexpr (PF) {
      // in flight across (a) and (b): the LDS-only barriers do not wait for them
      load_w(c0 + 16 < hp ? c0 + 16 : c0, win_n, wo_n, w1_n, w2_n);
    } else {
      stage_load(c0 + 16 < hp ? c0 + 16 : c0);     // next chunk's W_in rows; written to the other LDS buffer after (a)
      load_small(c0 + 16 < hp ? c0 + 16 : c0, wo_n, w1_n, w2_n);
    }

    UIR_ST(9);
    // ---- (a) t1 chunk on tile + ring ----
    if constexpr (SB) {
      // the chunk's W_in fragment is split once per 16-deep k step and meets every row tile of the wave
      f32x16 d[T::RTW];
#pragma unroll
      for (int q = 0; q < T::RTW; ++q)
#pragma unroll
        for (int i = 0; i < 16; ++i) d[q][i] = 0.f;
#pragma unroll
      for (int s = 0; s < C / 16; ++s) {
        UirBf3 wp;
        if constexpr (PF) {
          wp = uir_split8(win[2 * s], win[2 * s + 1]);
        } else if constexpr (!WPRE) {
          wp = uir_split8(*reinterpret_cast<const float4*>(Ws + wbuf * 32 * WS + li * WS + 16 * s + 4 * h2),
                          *reinterpret_cast<const float4*>(Ws + wbuf * 32 * WS + li * WS + 16 * s + 8 + 4 * h2));
        } else {
          const __bf16* src = Wp + wbuf * 3 * PLSZ + li * RSB + 16 * s + 8 * h2;
          wp.h = *reinterpret_cast<const uir_bf16x8*>(src);
          wp.m = *reinterpret_cast<const uir_bf16x8*>(src + PLSZ);
          wp.l = *reinterpret_cast<const uir_bf16x8*>(src + 2 * PLSZ);
        }
#pragma unroll
        for (int q = 0; q < T::RTW; ++q)
          if (wave + 4 * q < T::NRT1) d[q] = uir_mfma_split(Ap[q][s], wp, d[q]);
      }
      UIR_ST(10);
#pragma unroll
      for (int q = 0; q < T::RTW; ++q) {
        const int r = wave + 4 * q;
        if (r < T::NRT1) {
          float* dst = T1s + (r * 32 + 4 * h2) * T::S1 + li;
#pragma unroll
          for (int i = 0; i < 16; ++i) dst[((i & 3) + 8 * (i >> 2)) * T::S1] = d[q][i];
        }
      }
    } else {
#pragma unroll
    for (int q = 0; q < T::RTW; ++q) {
      const int r = wave + 4 * q;
      if (r < T::NRT1) {
        f32x16 d;
#pragma unroll
        for (int i = 0; i < 16; ++i) d[i] = 0.f;
#pragma unroll
        for (int jj = 0; jj < T::KJ; ++jj) {
          float4 wv_;
          if constexpr (PF) wv_ = win[jj];
          else wv_ = *reinterpret_cast<const float4*>(Ws + wbuf * 32 * WS + li * WS + 8 * jj + 4 * h2);
          d = uir_mfma(A[q][4 * jj + 0], wv_.x, d);
          d = uir_mfma(A[q][4 * jj + 1], wv_.y, d);
          d = uir_mfma(A[q][4 * jj + 2], wv_.z, d);
          d = uir_mfma(A[q][4 * jj + 3], wv_.w, d);
        }
        float* dst = T1s + (r * 32 + 4 * h2) * T::S1 + li;
#pragma unroll
        for (int i = 0; i < 16; ++i) dst[((i & 3) + 8 * (i >> 2)) * T::S1] = d[i];
      }
    }
    }
    UIR_ST(1);
    lds_barrier();
    UIR_ST(2);

    // The prefetched weights are needed only at the end of the chunk, but the compiler would wait for them with
    // vmcnt(0) AFTER this chunk's T1/T2 stores were issued (VMEM ops retire in issue order and it cannot count stores
    // issued under branches), exposing the full store latency every chunk (measured: +250 us per 1.6 GB).  Pin them
    // here instead: the wait lands before the stores, when the youngest outstanding store is a whole phase old.
    if constexpr (PF) {
#pragma unroll
      for (int jj = 0; jj < T::KJ; ++jj) { pin(win_n[jj].x); pin(win_n[jj].y); pin(win_n[jj].z); pin(win_n[jj].w); }
    } else {
      stage_store(wbuf ^ 1);     // (a) of chunk c-1 was the last reader of that buffer; visible after the next barrier
    }
    if constexpr (SB) {
#pragma unroll
      for (int t = 0; t < 8; ++t) pin(PF ? wo[t] : wo_n[t]);
      if constexpr (!TAPS_LDS) {
#pragma unroll
        for (int t = 0; t < 9; ++t) { pin(PF ? w1[t] : w1_n[t]); pin(PF ? w2[t] : w2_n[t]); }
      }
    } else {
#pragma unroll
    for (int t = 0; t < 8; ++t) pin(wo_n[t]);
#pragma unroll
    for (int t = 0; t < 9; ++t) { pin(w1_n[t]); pin(w2_n[t]); }
    }

    // ---- T1 write-out (training): interior rows of the LDS chunk as 16-byte stores, 8 lanes per pixel ----
    const unsigned cbytes = a.il ? 8u : 4u;      // bytes per hidden channel of the chunk offset (interleaved: 2 columns per channel)
    auto t1_store = [&](int it) {
      if (st_off[it] != ~0u) {
        const float4 v = *reinterpret_cast<const float4*>(T1s + st_lds[it]);
        st4(reinterpret_cast<char*>(a.T1) + (st_off[it] + (unsigned)c0 * cbytes), v);
      }
    };
    auto t2_store = [&](int it) {
      if (st_off[it] != ~0u) {
        const int e = tid + 256 * it;
        const float4 v = *reinterpret_cast<const float4*>(Gs + (e >> 3) * T::SG + 4 * (e & 7));
        st4(reinterpret_cast<char*>(a.T2) + (st_off[it] + (unsigned)c0 * cbytes), v);
      }
    };
    if (STORE && a.T1) {                         // T1 == NULL (uniform): the backward recomputes the pre-depthwise values (uir_dwln_bwd, XP == NULL)
#pragma unroll
      for (int it = 0; it < T::NST; ++it) t1_store(it);
    }

    UIR_ST(3);
    // ---- (b) depthwise 3x3: rolling three-row window (rotating register names, no shifts), row y+3
    //      loaded while row y is being computed; x1/x2 as packed pairs (v_pk_fma_f32) ----
    {
      const float* rp = T1s + (ys * T::RW + sx) * T::S1 + cp;
      f32x2 rw[3][3], wv[9];
      if constexpr (TAPS_LDS) {
        const float* wr = WT + (c0 + cp) * SWT;
        float w18[18];
        if constexpr (SWT == 20) {
          const float4 q0 = *reinterpret_cast<const float4*>(wr), q1 = *reinterpret_cast<const float4*>(wr + 4),
                       q2 = *reinterpret_cast<const float4*>(wr + 8), q3 = *reinterpret_cast<const float4*>(wr + 12);
          const float2 q4 = *reinterpret_cast<const float2*>(wr + 16);
          const float t18[18] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w, q2.x, q2.y, q2.z, q2.w, q3.x, q3.y, q3.z, q3.w, q4.x, q4.y};
#pragma unroll
          for (int t = 0; t < 18; ++t) w18[t] = t18[t];
        } else {
#pragma unroll
          for (int t = 0; t < 9; ++t) {
            const float2 q = *reinterpret_cast<const float2*>(wr + 2 * t);
            w18[2 * t] = q.x;
            w18[2 * t + 1] = q.y;
          }
        }
#pragma unroll
        for (int t = 0; t < 9; ++t) wv[t] = f32x2{w18[t], w18[9 + t]};
      } else {
#pragma unroll
        for (int t = 0; t < 9; ++t) wv[t] = f32x2{w1[t], w2[t]};
      }
#pragma unroll
      for (int rr = 0; rr < 3; ++rr)
#pragma unroll
        for (int dx = 0; dx < 3; ++dx)
          rw[rr][dx] = f32x2{rp[(rr * T::RW + dx) * T::S1], rp[(rr * T::RW + dx) * T::S1 + 16]};
      rp += 3 * T::RW * T::S1;
      float* gdst = Gs + (ys * TW + sx) * T::SG + cp;
      auto step = [&](f32x2 (&p0)[3], f32x2 (&p1)[3], f32x2 (&p2)[3]) {
        f32x2 t = wv[0] * p0[0];
        t = __builtin_elementwise_fma(wv[1], p0[1], t);
        t = __builtin_elementwise_fma(wv[2], p0[2], t);
        // p0 is free now: fetch ring row y+3 (the last fetches read past the ring: inside smem, never used)
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) p0[dx] = f32x2{rp[dx * T::S1], rp[dx * T::S1 + 16]};
        rp += T::RW * T::S1;
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
          t = __builtin_elementwise_fma(wv[3 + kx], p1[kx], t);
          t = __builtin_elementwise_fma(wv[6 + kx], p2[kx], t);
        }
        gdst[0] = t.x;
        gdst[16] = t.y;
        gdst += TW * T::SG;
      };
#pragma unroll 1
      for (int yb = 0; yb < T::ROWS; yb += 3) {
        step(rw[0], rw[1], rw[2]);
        if (yb + 1 < T::ROWS) step(rw[1], rw[2], rw[0]);
        if (yb + 2 < T::ROWS) step(rw[2], rw[0], rw[1]);
      }
    }
    UIR_ST(4);
    lds_barrier();
    UIR_ST(5);

    // ---- T2 write-out (training) ----
    if (STORE) {
#pragma unroll
      for (int it = 0; it < T::NST; ++it) t2_store(it);
    }

    UIR_ST(6);
    // ---- (c) y += (gelu(t2a) * t2b) . W_out^T over this chunk's 16 hidden channels ----
    UirBf3 wop;
    if constexpr (SB) wop = uir_split8(wo);
#pragma unroll
    for (int j = 0; j < T::UPW; ++j) {
      const int u = wave + 4 * j;
      if (u < T::UNITS) {
        const int rt = u / T::NT;
        const float* gr = Gs + (rt * 32 + li) * T::SG + 4 * h2;
        const float4 a0 = *reinterpret_cast<const float4*>(gr);
        const float4 a1 = *reinterpret_cast<const float4*>(gr + 8);
        const float4 b0 = *reinterpret_cast<const float4*>(gr + 16);
        const float4 b1 = *reinterpret_cast<const float4*>(gr + 24);
        if constexpr (SB) {
          const float gv[8] = {uir_gelu(a0.x) * b0.x, uir_gelu(a0.y) * b0.y, uir_gelu(a0.z) * b0.z, uir_gelu(a0.w) * b0.w,
                               uir_gelu(a1.x) * b1.x, uir_gelu(a1.y) * b1.y, uir_gelu(a1.z) * b1.z, uir_gelu(a1.w) * b1.w};
          acc[j] = uir_mfma_split(uir_split8(gv), wop, acc[j]);
          continue;
        }
        acc[j] = uir_mfma(uir_gelu(a0.x) * b0.x, wo[0], acc[j]);
        acc[j] = uir_mfma(uir_gelu(a0.y) * b0.y, wo[1], acc[j]);
        acc[j] = uir_mfma(uir_gelu(a0.z) * b0.z, wo[2], acc[j]);
        acc[j] = uir_mfma(uir_gelu(a0.w) * b0.w, wo[3], acc[j]);
        acc[j] = uir_mfma(uir_gelu(a1.x) * b1.x, wo[4], acc[j]);
        acc[j] = uir_mfma(uir_gelu(a1.y) * b1.y, wo[5], acc[j]);
        acc[j] = uir_mfma(uir_gelu(a1.z) * b1.z, wo[6], acc[j]);
        acc[j] = uir_mfma(uir_gelu(a1.w) * b1.w, wo[7], acc[j]);
      }
    }
    if constexpr (PF) {
#pragma unroll
      for (int jj = 0; jj < T::KJ; ++jj) win[jj] = win_n[jj];
    }
    if constexpr (!SB || !PF) {
#pragma unroll
      for (int t = 0; t < 8; ++t) wo[t] = wo_n[t];
    }
    if constexpr (!TAPS_LDS && (!SB || !PF)) {        // (SB at C = 128: taps in registers, fetched a chunk ahead with W_out's fragment)
#pragma unroll
      for (int t = 0; t < 9; ++t) { w1[t] = w1_n[t]; w2[t] = w2_n[t]; }
    }
    wbuf ^= 1;
    UIR_ST(7);
    // no barrier: (a) of the next chunk writes T1s (all reads of it finished before the barrier
    // above) and Gs is rewritten only after the next barrier, which every wave reaches after its (c)
  }

  // ---- epilogue: residual add (all loads first, clamped, no branches), predicated stores ----
#pragma unroll
  for (int j = 0; j < T::UPW; ++j) {
    const int u = wave + 4 * j;
    if (u < T::UNITS) {
      const int rt = u / T::NT;
      const int nn = my_n < C ? my_n : 0;
      long mm[16];
      float res[16];
      bool ok[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int p = rt * 32 + (i & 3) + 8 * (i >> 2) + 4 * h2;
        const int iy = y0 + p / TW, ix = x0 + p % TW;
        ok[i] = iy < a.H && ix < a.Wd && my_n < C;
        mm[i] = ok[i] ? img + (long)iy * a.Wd + ix : img;
        res[i] = a.X[mm[i] * a.ldx + nn];
      }
#pragma unroll
      for (int i = 0; i < 16; ++i)
        if (ok[i]) a.Y[mm[i] * a.ldy + nn] = acc[j][i] + res[i];
    }
  }
  UIR_ST(8);
  UIR_ST_FLUSH();
}

// ------------------------------------------------------------------------------------------------
// uir_qkv_fwd: norm1 -> qkv 1x1 (C -> 3C) -> depthwise 3x3 of an MDTA block in one pass over the pixels
// (SpectralTransformer.py:93-94 with the norm1 of :144).  Same tile / chunk skeleton as uir_gdfn_fwd without the gate
// and the second 1x1: chunks of 32 consecutive output channels, thread (cp, column) owns channels cp and cp + 16.
// Both results are written once: the pre-depthwise tensor (only the backward reads it: streaming stores) and qkv.
template <int C, int TH, int TW, bool SB>
__global__ __launch_bounds__(256, 2) void uir_qkv_fwd_kernel(const UirQkvFwdArgs a) {
  using T = GdfnTile<C, TH, TW>;
  extern __shared__ float smem[];
  float* T1s = smem;
  float* Gs = smem + T::NRT1 * 32 * T::S1;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, h2 = lane >> 5;
  const int tiles_x = (a.Wd + TW - 1) / TW, tiles_y = (a.H + TH - 1) / TH;
  int bid = blockIdx.x;
  const int tx = bid % tiles_x;
  bid /= tiles_x;
  const int ty = bid % tiles_y;
  const int b = bid / tiles_y;
  const int y0 = ty * TH, x0 = tx * TW;
  const long img = (long)b * a.H * a.Wd;
  constexpr int HD = 3 * C;
  // C <= 32, split-bf16 form: the depthwise taps of all 3C channels staged once per workgroup (see uir_gdfn_fwd_kernel); row of
  // pair (chunk, cp) = 9 taps of channel 32 chunk + cp | 9 of channel 32 chunk + 16 + cp, zeros past 3C
  constexpr bool TAPS_LDS = SB && C <= 32;
  constexpr int SWT = 20, NPR = ((HD + 31) / 32) * 16, NTAP = TAPS_LDS ? (NPR * 18 + 255) / 256 : 1;
  float* WT = smem + T::LDS_FLOATS;
  float tapv[NTAP];
  if constexpr (TAPS_LDS) {
#pragma unroll
    for (int it = 0; it < NTAP; ++it) {
      const int e = tid + 256 * it, pr = e / 18, q = e - 18 * pr;
      const int ch = 32 * (pr >> 4) + (pr & 15) + (q < 9 ? 0 : 16);
      tapv[it] = a.Wdw[(pr < NPR && ch < HD) ? ch * 9 + (q < 9 ? q : q - 9) : 0];
    }
  }

  // ---- phase 0: x on tile + ring -> LayerNorm -> A-operand registers ----
  float A[T::RTW][C / 2];
  {
    float4 gam[T::KJ], bet[T::KJ];
#pragma unroll
    for (int jj = 0; jj < T::KJ; ++jj) {
      gam[jj] = *reinterpret_cast<const float4*>(a.ln_g + 8 * jj + 4 * h2);
      bet[jj] = *reinterpret_cast<const float4*>(a.ln_b + 8 * jj + 4 * h2);
    }
#pragma unroll
    for (int q = 0; q < T::RTW; ++q) {
      const int r = wave + 4 * q;
      if (r < T::NRT1) {
        const int p = r * 32 + li;
        const int ry = p / T::RW, rx = p - ry * T::RW;
        const int iy = y0 + ry - 1, ix = x0 + rx - 1;
        const bool valid = p < T::NPX1 && iy >= 0 && iy < a.H && ix >= 0 && ix < a.Wd;
        const long m = valid ? img + (long)iy * a.Wd + ix : img;
        const float* xr = a.X + m * a.ldx + 4 * h2;
        float4 v[T::KJ];
        float s = 0.f;
#pragma unroll
        for (int jj = 0; jj < T::KJ; ++jj) v[jj] = *reinterpret_cast<const float4*>(xr + 8 * jj);
        __builtin_amdgcn_sched_barrier(0);          // all of the row's loads in flight before the first is summed (one variant waited for each)
#pragma unroll
        for (int jj = 0; jj < T::KJ; ++jj) s += (v[jj].x + v[jj].y) + (v[jj].z + v[jj].w);
        s += __shfl_xor(s, 32);
        const float mean = s / (float)C;
        float qq = 0.f;
#pragma unroll
        for (int jj = 0; jj < T::KJ; ++jj) {
          const float d0 = v[jj].x - mean, d1 = v[jj].y - mean, d2 = v[jj].z - mean, d3 = v[jj].w - mean;
          qq += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
        }
        qq += __shfl_xor(qq, 32);
        const float rstd = rsqrtf(qq / (float)C + a.ln_eps);
        const float keep = valid ? 1.f : 0.f;
#pragma unroll
        for (int jj = 0; jj < T::KJ; ++jj) {
          A[q][4 * jj + 0] = keep * fmaf((v[jj].x - mean) * rstd, gam[jj].x, bet[jj].x);
          A[q][4 * jj + 1] = keep * fmaf((v[jj].y - mean) * rstd, gam[jj].y, bet[jj].y);
          A[q][4 * jj + 2] = keep * fmaf((v[jj].z - mean) * rstd, gam[jj].z, bet[jj].z);
          A[q][4 * jj + 3] = keep * fmaf((v[jj].w - mean) * rstd, gam[jj].w, bet[jj].w);
        }
        const bool interior = ry >= 1 && ry <= TH && rx >= 1 && rx <= TW;
        if (a.stats && h2 == 0 && valid && interior)
          *reinterpret_cast<float2*>(a.stats + m * 2) = make_float2(mean, rstd);
      }
    }
  }

  if constexpr (TAPS_LDS) {                 // visible after chunk 0's first barrier
#pragma unroll
    for (int it = 0; it < NTAP; ++it) {
      const int e = tid + 256 * it, pr = e / 18, q = e - 18 * pr;
      const int ch = 32 * (pr >> 4) + (pr & 15) + (q < 9 ? 0 : 16);
      if (pr < NPR) WT[pr * SWT + q] = ch < HD ? tapv[it] : 0.f;
    }
  }
  UirBf3 Ap[SB ? T::RTW : 1][SB ? C / 16 : 1];       // SB: A operand as bf16 planes, split once per tile (uir_gdfn_fwd_kernel)
  if constexpr (SB) {
#pragma unroll
    for (int q = 0; q < T::RTW; ++q)
#pragma unroll
      for (int s = 0; s < C / 16; ++s) {
        const float x[8] = {A[q][8 * s], A[q][8 * s + 1], A[q][8 * s + 2], A[q][8 * s + 3], A[q][8 * s + 4], A[q][8 * s + 5], A[q][8 * s + 6], A[q][8 * s + 7]};
        Ap[q][s] = uir_split8(x);
      }
  }
  const int cp = tid & 15;
  const int strip = (tid >> 6) * 4 + (((lane >> 4) & 1) << 1 | (lane >> 5));
  const int sx = strip % TW, ys = (strip / TW) * T::ROWS;

  // write-out slots: float4 j (0..7) of pixel p covers chunk columns 4j..4j+3; byte offsets without the chunk term
  unsigned st_pre[T::NST], st_out[T::NST];
  int st_lds[T::NST];
#pragma unroll
  for (int it = 0; it < T::NST; ++it) {
    const int e = tid + 256 * it, p = e >> 3, j = e & 7;
    const int py = p / TW, px = p - py * TW;
    const int iy = y0 + py, ix = x0 + px;
    const bool ok = p < T::NI && iy < a.H && ix < a.Wd;
    const long m = img + (long)iy * a.Wd + ix;
    st_pre[it] = (ok && a.Ypre) ? (unsigned)((m * a.ldp + 4 * j) * 4) : ~0u;     // Ypre == NULL: not stored (uir_dwln_bwd recomputes it)
    st_out[it] = ok ? (unsigned)((m * a.ldq + 4 * j) * 4) : ~0u;
    st_lds[it] = ((py + 1) * T::RW + px + 1) * T::S1 + 4 * j;
  }

  // wide levels, split-bf16 form: the chunk's 32 rows of W_qkv go through LDS as three bf16 planes, split ONCE by the staging threads and
  // fetched a chunk ahead (uir_gdfn_fwd_kernel's scheme).  Before, every lane loaded its half row (C / 2 floats) at the top of the chunk
  // and consumed it at once - the load's latency and 80 split instructions per k step and wave, every chunk
  constexpr bool WPRE = SB && C > 32;
  constexpr int RSB = C + 8, PLSZ = 32 * RSB;           // plane row stride / plane size (bf16 units)
  constexpr int NSTG = WPRE ? (32 * (C / 4)) / 256 : 1;
  __bf16* Wp = reinterpret_cast<__bf16*>(smem + T::LDS_FLOATS);
  float4 stg[NSTG];
  auto stage_load = [&](int c0) {
#pragma unroll
    for (int it = 0; it < NSTG; ++it) {
      const int e = tid + 256 * it, r = e / (C / 4), q = e - r * (C / 4);
      const bool ok = c0 + r < HD;
      stg[it] = keep4(*reinterpret_cast<const float4*>(a.Wqkv + (long)(ok ? c0 + r : 0) * C + 4 * q), ok);
    }
  };
  auto stage_store = [&](int buf) {
#pragma unroll
    for (int it = 0; it < NSTG; ++it) {
      const int e = tid + 256 * it, r = e / (C / 4), q = e - r * (C / 4);
      const UirBf3x4 pl = uir_split4(stg[it]);
      const int slot = 16 * (q >> 2) + 8 * (q & 1) + 4 * ((q >> 1) & 1);     // k = 16 s + 8 g + 4 h -> slot 16 s + 8 h + 4 g
      __bf16* d = Wp + buf * 3 * PLSZ + r * RSB + slot;
      *reinterpret_cast<uint2*>(d) = pl.h;
      *reinterpret_cast<uint2*>(d + PLSZ) = pl.m;
      *reinterpret_cast<uint2*>(d + 2 * PLSZ) = pl.l;
    }
  };
  auto load_w = [&](int c0, float4 (&win)[T::KJ], float (&w1)[9], float (&w2)[9]) {
    if constexpr (!WPRE) {
      const int cc = c0 + li;
      const bool ok = cc < HD;
      const float* wr = a.Wqkv + (long)(ok ? cc : 0) * C + 4 * h2;
#pragma unroll
      for (int jj = 0; jj < T::KJ; ++jj) win[jj] = keep4(*reinterpret_cast<const float4*>(wr + 8 * jj), ok);
    }
    if constexpr (!TAPS_LDS) {
      const int c1 = c0 + cp, c2 = c0 + 16 + cp;
      const bool ok1 = c1 < HD, ok2 = c2 < HD;
      const float* p1 = a.Wdw + (long)(ok1 ? c1 : 0) * 9;
      const float* p2 = a.Wdw + (long)(ok2 ? c2 : 0) * 9;
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        const float u1 = p1[t], u2 = p2[t];
        w1[t] = ok1 ? u1 : 0.f;
        w2[t] = ok2 ? u2 : 0.f;
      }
    }
  };
  constexpr bool PF = C <= 32;
  constexpr int NPF = PF ? T::KJ : 1;
  float4 win[T::KJ], win_n[NPF];
  float w1[9], w2[9], w1_n[9], w2_n[9];
  if (PF) load_w(0, win, w1, w2);
  int wbuf = 0;
  if constexpr (WPRE) {
    stage_load(0);
    stage_store(0);
    lds_barrier();
  }

  // ---- optional Gram of the attention logits (C <= 32, a.gram_slab): the q and k tiles pass through Gs anyway.  Each wave contracts its
  // quarter of the tile's pixels on the f32 MFMA (A = q^T: lane = q channel, B = k: lane = k channel, k-slot h2 = pixel 2 s + h2); at
  // C = 16 q and k share chunk 0 (columns 0..15 | 16..31) and the chunk's self-product holds q k^T off its diagonal blocks and both
  // norms on its diagonal.  Pixels outside the image (ragged last tile row / column) do not count.
  constexpr bool GRAM_OK = C <= 32;
  constexpr int GSTEPS = GRAM_OK ? T::NI / 8 : 1;            // 2 pixels per step, a quarter of the tile per wave
  const bool gram = GRAM_OK && a.gram_slab != nullptr;
  float qa[GSTEPS];
  f32x16 gd;
  float sq_q = 0.f, sq_k = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) gd[i] = 0.f;
  auto gram_frag = [&](float (&o)[GSTEPS]) {
#pragma unroll
    for (int s = 0; s < GSTEPS; ++s) {
      const int p = (T::NI / 4) * wave + 2 * s + h2;
      const int py = p / TW, px = p - py * TW;
      const float v = Gs[p * T::SG + li];
      o[s] = (y0 + py < a.H && x0 + px < a.Wd) ? v : 0.f;
    }
  };

  for (int c0 = 0; c0 < HD; c0 += 32) {
    if constexpr (PF) {
      load_w(c0 + 32 < HD ? c0 + 32 : c0, win_n, w1_n, w2_n);
    } else {
      if constexpr (WPRE) stage_load(c0 + 32 < HD ? c0 + 32 : c0);     // next chunk's rows: into the other buffer after (a)
      load_w(c0, win, w1, w2);                                          // (WPRE: the depthwise taps only)
    }
    // ---- (a) 32 output channels of the 1x1 on tile + ring ----
    if constexpr (SB) {
      f32x16 d[T::RTW];
#pragma unroll
      for (int q = 0; q < T::RTW; ++q)
#pragma unroll
        for (int i = 0; i < 16; ++i) d[q][i] = 0.f;
#pragma unroll
      for (int s = 0; s < C / 16; ++s) {
        UirBf3 wp;
        if constexpr (WPRE) {
          const __bf16* src = Wp + wbuf * 3 * PLSZ + li * RSB + 16 * s + 8 * h2;
          wp.h = *reinterpret_cast<const uir_bf16x8*>(src);
          wp.m = *reinterpret_cast<const uir_bf16x8*>(src + PLSZ);
          wp.l = *reinterpret_cast<const uir_bf16x8*>(src + 2 * PLSZ);
        } else {
          wp = uir_split8(win[2 * s], win[2 * s + 1]);
        }
#pragma unroll
        for (int q = 0; q < T::RTW; ++q)
          if (wave + 4 * q < T::NRT1) d[q] = uir_mfma_split(Ap[q][s], wp, d[q]);
      }
#pragma unroll
      for (int q = 0; q < T::RTW; ++q) {
        const int r = wave + 4 * q;
        if (r < T::NRT1) {
          float* dst = T1s + (r * 32 + 4 * h2) * T::S1 + li;
#pragma unroll
          for (int i = 0; i < 16; ++i) dst[((i & 3) + 8 * (i >> 2)) * T::S1] = d[q][i];
        }
      }
    } else {
#pragma unroll
    for (int q = 0; q < T::RTW; ++q) {
      const int r = wave + 4 * q;
      if (r < T::NRT1) {
        f32x16 d;
#pragma unroll
        for (int i = 0; i < 16; ++i) d[i] = 0.f;
#pragma unroll
        for (int jj = 0; jj < T::KJ; ++jj) {
          d = uir_mfma(A[q][4 * jj + 0], win[jj].x, d);
          d = uir_mfma(A[q][4 * jj + 1], win[jj].y, d);
          d = uir_mfma(A[q][4 * jj + 2], win[jj].z, d);
          d = uir_mfma(A[q][4 * jj + 3], win[jj].w, d);
        }
        float* dst = T1s + (r * 32 + 4 * h2) * T::S1 + li;
#pragma unroll
        for (int i = 0; i < 16; ++i) dst[((i & 3) + 8 * (i >> 2)) * T::S1] = d[i];
      }
    }
    }
    lds_barrier();
    if constexpr (WPRE) {
      stage_store(wbuf ^ 1);     // (a) of the previous chunk was the last reader of that buffer; visible after the next barrier
      wbuf ^= 1;                 // (this chunk's (a) is done: the next (a) reads what was just stored)
    }
    if constexpr (PF) {   // see uir_gdfn_fwd_kernel: the prefetched weights must be complete before this chunk's stores
#pragma unroll
      for (int jj = 0; jj < T::KJ; ++jj) { pin(win_n[jj].x); pin(win_n[jj].y); pin(win_n[jj].z); pin(win_n[jj].w); }
      if constexpr (!TAPS_LDS) {
#pragma unroll
        for (int t = 0; t < 9; ++t) { pin(w1_n[t]); pin(w2_n[t]); }
      }
    }
    // ---- pre-depthwise write-out (read again only by the backward: streaming stores) ----
#pragma unroll
    for (int it = 0; it < T::NST; ++it)
      if (st_pre[it] != ~0u && c0 + 4 * ((tid + 256 * it) & 7) < HD) {
        const float4 v = *reinterpret_cast<const float4*>(T1s + st_lds[it]);
        st4(reinterpret_cast<char*>(a.Ypre) + (st_pre[it] + (unsigned)c0 * 4u), v);
      }
    // ---- (b) depthwise 3x3 ----
    {
      const float* rp = T1s + (ys * T::RW + sx) * T::S1 + cp;
      f32x2 rw[3][3], wv[9];
      if constexpr (TAPS_LDS) {
        const float* wr = WT + ((c0 >> 5) * 16 + cp) * SWT;
        const float4 q0 = *reinterpret_cast<const float4*>(wr), q1 = *reinterpret_cast<const float4*>(wr + 4),
                     q2 = *reinterpret_cast<const float4*>(wr + 8), q3 = *reinterpret_cast<const float4*>(wr + 12);
        const float2 q4 = *reinterpret_cast<const float2*>(wr + 16);
        const float w18[18] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w, q2.x, q2.y, q2.z, q2.w, q3.x, q3.y, q3.z, q3.w, q4.x, q4.y};
#pragma unroll
        for (int t = 0; t < 9; ++t) wv[t] = f32x2{w18[t], w18[9 + t]};
      } else {
#pragma unroll
        for (int t = 0; t < 9; ++t) wv[t] = f32x2{w1[t], w2[t]};
      }
#pragma unroll
      for (int rr = 0; rr < 3; ++rr)
#pragma unroll
        for (int dx = 0; dx < 3; ++dx)
          rw[rr][dx] = f32x2{rp[(rr * T::RW + dx) * T::S1], rp[(rr * T::RW + dx) * T::S1 + 16]};
      rp += 3 * T::RW * T::S1;
      float* gdst = Gs + (ys * TW + sx) * T::SG + cp;
      auto step = [&](f32x2 (&p0)[3], f32x2 (&p1)[3], f32x2 (&p2)[3]) {
        f32x2 t = wv[0] * p0[0];
        t = __builtin_elementwise_fma(wv[1], p0[1], t);
        t = __builtin_elementwise_fma(wv[2], p0[2], t);
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) p0[dx] = f32x2{rp[dx * T::S1], rp[dx * T::S1 + 16]};
        rp += T::RW * T::S1;
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
          t = __builtin_elementwise_fma(wv[3 + kx], p1[kx], t);
          t = __builtin_elementwise_fma(wv[6 + kx], p2[kx], t);
        }
        gdst[0] = t.x;
        gdst[16] = t.y;
        gdst += TW * T::SG;
      };
#pragma unroll 1
      for (int yb = 0; yb < T::ROWS; yb += 3) {
        step(rw[0], rw[1], rw[2]);
        if (yb + 1 < T::ROWS) step(rw[1], rw[2], rw[0]);
        if (yb + 2 < T::ROWS) step(rw[2], rw[0], rw[1]);
      }
    }
    lds_barrier();
    // ---- qkv write-out ----
#pragma unroll
    for (int it = 0; it < T::NST; ++it)
      if (st_out[it] != ~0u && c0 + 4 * ((tid + 256 * it) & 7) < HD) {
        const int e = tid + 256 * it;
        const float4 v = *reinterpret_cast<const float4*>(Gs + (e >> 3) * T::SG + 4 * (e & 7));
        *reinterpret_cast<float4*>(reinterpret_cast<char*>(a.Y) + (st_out[it] + (unsigned)c0 * 4u)) = v;
      }
    if constexpr (GRAM_OK) {
      // Gs still holds this chunk's tile: it is rewritten only behind the next chunk's first barrier, which this wave has not reached
      if (gram) {
        if (C == 16 && c0 == 0) {
          gram_frag(qa);
#pragma unroll
          for (int s = 0; s < GSTEPS; ++s) gd = uir_mfma(qa[s], qa[s], gd);
        } else if (C == 32 && c0 == 0) {
          gram_frag(qa);
#pragma unroll
          for (int s = 0; s < GSTEPS; ++s) sq_q = fmaf(qa[s], qa[s], sq_q);
        } else if (C == 32 && c0 == 32) {
          float kb[GSTEPS];
          gram_frag(kb);
#pragma unroll
          for (int s = 0; s < GSTEPS; ++s) {
            gd = uir_mfma(qa[s], kb[s], gd);
            sq_k = fmaf(kb[s], kb[s], sq_k);
          }
        }
      }
    }
    if constexpr (PF) {
#pragma unroll
      for (int jj = 0; jj < T::KJ; ++jj) win[jj] = win_n[jj];
      if constexpr (!TAPS_LDS) {
#pragma unroll
        for (int t = 0; t < 9; ++t) { w1[t] = w1_n[t]; w2[t] = w2_n[t]; }
      }
    }
    // the next (a) writes T1s (all reads done before the barrier above); Gs is rewritten after the next barrier,
    // which every wave reaches only after its qkv write-out
  }
  if constexpr (GRAM_OK) {
    if (gram) {
      // the four waves' partials meet in T1s (dead now) in a fixed order; one slab row per tile: G [C][C] | sum q^2 [C] | sum k^2 [C]
      constexpr int RS = 1024 + 64;
      sq_q += __shfl_xor(sq_q, 32);
      sq_k += __shfl_xor(sq_k, 32);
      lds_barrier();
      float* red = T1s;
#pragma unroll
      for (int i = 0; i < 16; ++i) red[wave * RS + i * 64 + lane] = gd[i];
      if (h2 == 0) {
        red[wave * RS + 1024 + li] = sq_q;
        red[wave * RS + 1056 + li] = sq_k;
      }
      lds_barrier();
      float* row = a.gram_slab + (long)blockIdx.x * a.gram_stride;
      for (int e = tid; e < 1024; e += 256) {
        const float v = (red[e] + red[RS + e]) + (red[2 * RS + e] + red[3 * RS + e]);
        const int i = e >> 6, ln = e & 63;
        const int m = uir_mfma_row(i, ln >> 5), n = ln & 31;
        if (C == 32) {
          row[m * 32 + n] = v;
        } else {
          if (m < 16 && n >= 16) row[m * 16 + (n - 16)] = v;
          if (m == n) row[256 + m] = v;                     // sum q^2 at 256 + m (m < 16), sum k^2 at 272 + (m - 16)
        }
      }
      if (C == 32 && tid < 64) {
        const int e = 1024 + tid;
        row[e] = (red[e] + red[RS + e]) + (red[2 * RS + e] + red[3 * RS + e]);
      }
    }
  }
}

template <int C, int TH, int TW>
int qkv_fwd_launch(const UirQkvFwdArgs& a, hipStream_t st) {
  using T = GdfnTile<C, TH, TW>;
  const int tiles = a.B * ((a.H + TH - 1) / TH) * ((a.Wd + TW - 1) / TW);
  // + the staged depthwise taps (C <= 32, split-bf16 form): 16 rows of 20 floats per 32-channel chunk
  // ... / the two staged weight chunks as bf16 planes (wide levels)
  if (uir_fused_split())
    hipLaunchKernelGGL((uir_qkv_fwd_kernel<C, TH, TW, true>), dim3(tiles), dim3(256),
                       (size_t)T::LDS_FLOATS * 4 + (C <= 32 ? (size_t)((3 * C + 31) / 32) * 16 * 20 * 4 : (size_t)2 * 3 * 32 * (C + 8) * 2), st, a);
  else hipLaunchKernelGGL((uir_qkv_fwd_kernel<C, TH, TW, false>), dim3(tiles), dim3(256), (size_t)T::LDS_FLOATS * 4, st, a);
  return uir_launch_status();
}

// ------------------------------------------------------------------------------------------------
// uir_attnv_fwd: the pixel-sized tail of MDTA in one pass over the pixels (SpectralTransformer.py:100,107,110,113 after the
// algebraic folding of DESIGN.md section 5):  u = v . WA[b]^T,  vf = dw3x3(u),  x1 = vf . WB[b]^T + x
// with the per-sample C x C matrices WA, WB from uir_attn_fwd.  Same skeleton as uir_gdfn_fwd (no LayerNorm, no gate):
// chunks of 32 channels of u / vf; both are written once (the backward reads them).
template <int C, int TH, int TW, bool SB>
__global__ __launch_bounds__(256, 2) void uir_attnv_fwd_kernel(const UirAttnvFwdArgs a) {
  using T = GdfnTile<C, TH, TW>;
  extern __shared__ float smem[];
  float* T1s = smem;
  float* Gs = smem + T::NRT1 * 32 * T::S1;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, h2 = lane >> 5;
  const int tiles_x = (a.Wd + TW - 1) / TW, tiles_y = (a.H + TH - 1) / TH;
  int bid = blockIdx.x;
  const int tx = bid % tiles_x;
  bid /= tiles_x;
  const int ty = bid % tiles_y;
  const int b = bid / tiles_y;
  const int y0 = ty * TH, x0 = tx * TW;
  const long img = (long)b * a.H * a.Wd;
  const float* WA = a.WA + (long)b * C * C;
  const float* WB = a.WB + (long)b * C * C;

  // ---- phase 0: v on tile + ring in A-operand registers, zero outside the image ----
  float A[T::RTW][C / 2];
#pragma unroll
  for (int q = 0; q < T::RTW; ++q) {
    const int r = wave + 4 * q;
    if (r < T::NRT1) {
      const int p = r * 32 + li;
      const int ry = p / T::RW, rx = p - ry * T::RW;
      const int iy = y0 + ry - 1, ix = x0 + rx - 1;
      const bool valid = p < T::NPX1 && iy >= 0 && iy < a.H && ix >= 0 && ix < a.Wd;
      const long m = valid ? img + (long)iy * a.Wd + ix : img;
      const float* vr = a.V + m * a.ldv + 4 * h2;
      const float keep = valid ? 1.f : 0.f;
#pragma unroll
      for (int jj = 0; jj < T::KJ; ++jj) {
        const float4 v = *reinterpret_cast<const float4*>(vr + 8 * jj);
        A[q][4 * jj + 0] = keep * v.x;
        A[q][4 * jj + 1] = keep * v.y;
        A[q][4 * jj + 2] = keep * v.z;
        A[q][4 * jj + 3] = keep * v.w;
      }
    }
  }
  UirBf3 Ap[SB ? T::RTW : 1][SB ? C / 16 : 1];       // SB: A operand as bf16 planes, split once per tile (uir_gdfn_fwd_kernel)
  if constexpr (SB) {
#pragma unroll
    for (int q = 0; q < T::RTW; ++q)
#pragma unroll
      for (int s = 0; s < C / 16; ++s) {
        const float x[8] = {A[q][8 * s], A[q][8 * s + 1], A[q][8 * s + 2], A[q][8 * s + 3], A[q][8 * s + 4], A[q][8 * s + 5], A[q][8 * s + 6], A[q][8 * s + 7]};
        Ap[q][s] = uir_split8(x);
      }
  }
  f32x16 acc[T::UPW];
#pragma unroll
  for (int j = 0; j < T::UPW; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  const int my_n = (wave % T::NT) * 32 + li;

  const int cp = tid & 15;
  const int strip = (tid >> 6) * 4 + (((lane >> 4) & 1) << 1 | (lane >> 5));
  const int sx = strip % TW, ys = (strip / TW) * T::ROWS;

  unsigned st_u[T::NST], st_vf[T::NST];
  int st_lds[T::NST];
#pragma unroll
  for (int it = 0; it < T::NST; ++it) {
    const int e = tid + 256 * it, p = e >> 3, j = e & 7;
    const int py = p / TW, px = p - py * TW;
    const int iy = y0 + py, ix = x0 + px;
    const bool ok = p < T::NI && iy < a.H && ix < a.Wd;
    const long m = img + (long)iy * a.Wd + ix;
    st_u[it] = ok ? (unsigned)((m * a.ldu + 4 * j) * 4) : ~0u;
    st_vf[it] = ok ? (unsigned)((m * a.ldf + 4 * j) * 4) : ~0u;
    st_lds[it] = ((py + 1) * T::RW + px + 1) * T::S1 + 4 * j;
  }

  for (int c0 = 0; c0 < C; c0 += 32) {
    // per-chunk weights (small: C <= 128, L2-resident per sample)
    float4 wa[T::KJ];
    {
      const int cc = c0 + li;
      const bool ok = cc < C;
      const float* wr = WA + (long)(ok ? cc : 0) * C + 4 * h2;
#pragma unroll
      for (int jj = 0; jj < T::KJ; ++jj) wa[jj] = keep4(*reinterpret_cast<const float4*>(wr + 8 * jj), ok);
    }
    float4 wb[4];   // (c) B operand: WB[my_n][c0 + 8 jj + 4 h2 + e]
    {
      const bool nok = my_n < C;
      const float* wr = WB + (long)(nok ? my_n : 0) * C + c0 + 4 * h2;
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) {
        const bool ok = nok && (c0 + 8 * jj + 4 * h2) < C;
        wb[jj] = keep4(*reinterpret_cast<const float4*>(wr + (ok ? 8 * jj : 0)), ok);
      }
    }
    f32x2 wv[9];
    {
      const int c1 = c0 + cp, c2 = c0 + 16 + cp;
      const bool ok1 = c1 < C, ok2 = c2 < C;
      const float* p1 = a.Wdw + (long)(ok1 ? c1 : 0) * 9;
      const float* p2 = a.Wdw + (long)(ok2 ? c2 : 0) * 9;
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        const float u1 = p1[t], u2 = p2[t];
        wv[t] = f32x2{ok1 ? u1 : 0.f, ok2 ? u2 : 0.f};
      }
    }
    // ---- (a) u chunk on tile + ring ----
    if constexpr (SB) {
      f32x16 d[T::RTW];
#pragma unroll
      for (int q = 0; q < T::RTW; ++q)
#pragma unroll
        for (int i = 0; i < 16; ++i) d[q][i] = 0.f;
#pragma unroll
      for (int s = 0; s < C / 16; ++s) {
        const UirBf3 wp = uir_split8(wa[2 * s], wa[2 * s + 1]);
#pragma unroll
        for (int q = 0; q < T::RTW; ++q)
          if (wave + 4 * q < T::NRT1) d[q] = uir_mfma_split(Ap[q][s], wp, d[q]);
      }
#pragma unroll
      for (int q = 0; q < T::RTW; ++q) {
        const int r = wave + 4 * q;
        if (r < T::NRT1) {
          float* dst = T1s + (r * 32 + 4 * h2) * T::S1 + li;
#pragma unroll
          for (int i = 0; i < 16; ++i) dst[((i & 3) + 8 * (i >> 2)) * T::S1] = d[q][i];
        }
      }
    } else {
#pragma unroll
    for (int q = 0; q < T::RTW; ++q) {
      const int r = wave + 4 * q;
      if (r < T::NRT1) {
        f32x16 d;
#pragma unroll
        for (int i = 0; i < 16; ++i) d[i] = 0.f;
#pragma unroll
        for (int jj = 0; jj < T::KJ; ++jj) {
          d = uir_mfma(A[q][4 * jj + 0], wa[jj].x, d);
          d = uir_mfma(A[q][4 * jj + 1], wa[jj].y, d);
          d = uir_mfma(A[q][4 * jj + 2], wa[jj].z, d);
          d = uir_mfma(A[q][4 * jj + 3], wa[jj].w, d);
        }
        float* dst = T1s + (r * 32 + 4 * h2) * T::S1 + li;
#pragma unroll
        for (int i = 0; i < 16; ++i) dst[((i & 3) + 8 * (i >> 2)) * T::S1] = d[i];
      }
    }
    }
    lds_barrier();
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) { pin(wb[jj].x); pin(wb[jj].y); pin(wb[jj].z); pin(wb[jj].w); }
    // ---- u write-out ----
#pragma unroll
    for (int it = 0; it < T::NST; ++it)
      if (st_u[it] != ~0u && c0 + 4 * ((tid + 256 * it) & 7) < C) {
        const float4 v = *reinterpret_cast<const float4*>(T1s + st_lds[it]);
        *reinterpret_cast<float4*>(reinterpret_cast<char*>(a.U) + (st_u[it] + (unsigned)c0 * 4u)) = v;
      }
    // ---- (b) depthwise 3x3 ----
    {
      const float* rp = T1s + (ys * T::RW + sx) * T::S1 + cp;
      f32x2 rw[3][3];
#pragma unroll
      for (int rr = 0; rr < 3; ++rr)
#pragma unroll
        for (int dx = 0; dx < 3; ++dx)
          rw[rr][dx] = f32x2{rp[(rr * T::RW + dx) * T::S1], rp[(rr * T::RW + dx) * T::S1 + 16]};
      rp += 3 * T::RW * T::S1;
      float* gdst = Gs + (ys * TW + sx) * T::SG + cp;
      auto step = [&](f32x2 (&p0)[3], f32x2 (&p1)[3], f32x2 (&p2)[3]) {
        f32x2 t = wv[0] * p0[0];
        t = __builtin_elementwise_fma(wv[1], p0[1], t);
        t = __builtin_elementwise_fma(wv[2], p0[2], t);
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) p0[dx] = f32x2{rp[dx * T::S1], rp[dx * T::S1 + 16]};
        rp += T::RW * T::S1;
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
          t = __builtin_elementwise_fma(wv[3 + kx], p1[kx], t);
          t = __builtin_elementwise_fma(wv[6 + kx], p2[kx], t);
        }
        gdst[0] = t.x;
        gdst[16] = t.y;
        gdst += TW * T::SG;
      };
#pragma unroll 1
      for (int yb = 0; yb < T::ROWS; yb += 3) {
        step(rw[0], rw[1], rw[2]);
        if (yb + 1 < T::ROWS) step(rw[1], rw[2], rw[0]);
        if (yb + 2 < T::ROWS) step(rw[2], rw[0], rw[1]);
      }
    }
    lds_barrier();
    // ---- vf write-out ----
#pragma unroll
    for (int it = 0; it < T::NST; ++it)
      if (st_vf[it] != ~0u && c0 + 4 * ((tid + 256 * it) & 7) < C) {
        const int e = tid + 256 * it;
        const float4 v = *reinterpret_cast<const float4*>(Gs + (e >> 3) * T::SG + 4 * (e & 7));
        *reinterpret_cast<float4*>(reinterpret_cast<char*>(a.VF) + (st_vf[it] + (unsigned)c0 * 4u)) = v;
      }
    // ---- (c) x1 += vf chunk . WB^T ----
    UirBf3 wbp[SB ? 2 : 1];
    if constexpr (SB) { wbp[0] = uir_split8(wb[0], wb[1]); wbp[1] = uir_split8(wb[2], wb[3]); }
#pragma unroll
    for (int j = 0; j < T::UPW; ++j) {
      const int u = wave + 4 * j;
      if (u < T::UNITS) {
        const int rt = u / T::NT;
        const float* gr = Gs + (rt * 32 + li) * T::SG + 4 * h2;
        if constexpr (SB) {
#pragma unroll
          for (int s = 0; s < 2; ++s)
            acc[j] = uir_mfma_split(uir_split8(*reinterpret_cast<const float4*>(gr + 16 * s), *reinterpret_cast<const float4*>(gr + 16 * s + 8)), wbp[s], acc[j]);
          continue;
        }
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
          const float4 v = *reinterpret_cast<const float4*>(gr + 8 * jj);
          acc[j] = uir_mfma(v.x, wb[jj].x, acc[j]);
          acc[j] = uir_mfma(v.y, wb[jj].y, acc[j]);
          acc[j] = uir_mfma(v.z, wb[jj].z, acc[j]);
          acc[j] = uir_mfma(v.w, wb[jj].w, acc[j]);
        }
      }
    }
  }

  // ---- epilogue: + residual (block input), store ----
#pragma unroll
  for (int j = 0; j < T::UPW; ++j) {
    const int u = wave + 4 * j;
    if (u < T::UNITS) {
      const int rt = u / T::NT;
      const int nn = my_n < C ? my_n : 0;
      long mm[16];
      float res[16];
      bool ok[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int p = rt * 32 + (i & 3) + 8 * (i >> 2) + 4 * h2;
        const int iy = y0 + p / TW, ix = x0 + p % TW;
        ok[i] = iy < a.H && ix < a.Wd && my_n < C;
        mm[i] = ok[i] ? img + (long)iy * a.Wd + ix : img;
        res[i] = a.R[mm[i] * a.ldr + nn];
      }
#pragma unroll
      for (int i = 0; i < 16; ++i)
        if (ok[i]) a.Y[mm[i] * a.ldy + nn] = acc[j][i] + res[i];
    }
  }
}

template <int C, int TH, int TW>
int attnv_fwd_launch(const UirAttnvFwdArgs& a, hipStream_t st) {
  using T = GdfnTile<C, TH, TW>;
  const int tiles = a.B * ((a.H + TH - 1) / TH) * ((a.Wd + TW - 1) / TW);
  if (uir_fused_split()) hipLaunchKernelGGL((uir_attnv_fwd_kernel<C, TH, TW, true>), dim3(tiles), dim3(256), (size_t)T::LDS_FLOATS * 4, st, a);
  else hipLaunchKernelGGL((uir_attnv_fwd_kernel<C, TH, TW, false>), dim3(tiles), dim3(256), (size_t)T::LDS_FLOATS * 4, st, a);
  return uir_launch_status();
}

template <int C, int TH, int TW>
int gdfn_fwd_launch(const UirGdfnFwdArgs& a, hipStream_t st) {
  using T = GdfnTile<C, TH, TW>;
  const int tiles = a.B * ((a.H + TH - 1) / TH) * ((a.Wd + TW - 1) / TW);
  // + two staged W_in chunks (wide levels) + the staged depthwise taps (split-bf16 form).  A hidden width whose taps do not fit
  // beside two workgroups' tile buffers (or the C <= 32 form's 7 tap slots per thread) runs the f32-MFMA form, which reads them from global
  bool sb = uir_fused_split() != 0;
  // C <= 32: + the staged depthwise taps (split-bf16 form; hp <= 96: 7 tap slots per thread).  C = 64: + two staged f32 W_in chunks [32][C + 4]
  // + (split-bf16 form) the taps.  C = 128: + two staged W_in chunks, f32 or (split-bf16 form) three bf16 planes [32][C + 8] each.
  // A hidden width whose taps do not fit runs the f32-MFMA form, which reads them from global
  size_t lds = (size_t)T::LDS_FLOATS * 4;
  if (C <= 32) {
    if (sb && a.hp > 96) sb = false;
    if (sb) lds += (size_t)a.hp * 20 * 4;
  } else if (C == 64) {
    lds += (size_t)2 * 32 * (C + 4) * 4;
    if (sb && lds + (size_t)a.hp * 18 * 4 > 80 * 1024) sb = false;
    if (sb) lds += (size_t)a.hp * 18 * 4;
  } else {
    lds += sb ? (size_t)2 * 3 * 32 * (C + 8) * 2 : (size_t)2 * 32 * (C + 4) * 4;
  }
  if (a.T2) { if (sb) hipLaunchKernelGGL((uir_gdfn_fwd_kernel<C, TH, TW, true, true>), dim3(tiles), dim3(256), lds, st, a);
              else hipLaunchKernelGGL((uir_gdfn_fwd_kernel<C, TH, TW, true, false>), dim3(tiles), dim3(256), lds, st, a); }
  else { if (sb) hipLaunchKernelGGL((uir_gdfn_fwd_kernel<C, TH, TW, false, true>), dim3(tiles), dim3(256), lds, st, a);
         else hipLaunchKernelGGL((uir_gdfn_fwd_kernel<C, TH, TW, false, false>), dim3(tiles), dim3(256), lds, st, a); }
  return uir_launch_status();
}

}  // namespace
UIR_ST_EXPORT

extern "C" int uir_gdfn_fwd(const UirGdfnFwdArgs* p, void* stream) {
  const UirGdfnFwdArgs& a = *p;
  UIR_REQUIRE(a.X && a.ln_g && a.ln_b && a.Win && a.Wdw && a.Wout && a.Y);
  UIR_REQUIRE((a.T1 == nullptr || a.T2 != nullptr) && (a.il == 0 || a.il == 1));      // T2 alone: the backward recomputes T1 (C <= 32, uir_dwln_bwd)
  UIR_REQUIRE(a.B > 0 && a.H > 0 && a.Wd > 0 && a.h > 0 && a.hp >= a.h && (a.hp & 15) == 0);
  UIR_REQUIRE(a.ldx >= a.C && (a.ldx & 3) == 0 && a.ldy >= a.C && uir_aligned16(a.X));
  UIR_REQUIRE(uir_aligned16(a.ln_g) && uir_aligned16(a.ln_b) && uir_aligned16(a.Win));
  UIR_REQUIRE(!a.T2 || (a.ldt >= 2 * a.hp && (a.ldt & 3) == 0 && uir_aligned16(a.T1) && uir_aligned16(a.T2) &&
                        (long)a.B * a.H * a.Wd * a.ldt < (1L << 30)));
  UIR_REQUIRE((long)a.B * ((a.H + 7) / 8) * ((a.Wd + 7) / 8) < (1L << 30));
  hipStream_t st = static_cast<hipStream_t>(stream);
  switch (a.C) {
    case 16: return gdfn_fwd_launch<16, 12, 16>(a, st);
    case 32: return gdfn_fwd_launch<32, 12, 16>(a, st);
    case 64: return gdfn_fwd_launch<64, 8, 16>(a, st);
    case 128: return gdfn_fwd_launch<128, 4, 8>(a, st);
    default: return UIR_EINVAL;
  }
}

extern "C" int uir_qkv_fwd(const UirQkvFwdArgs* p, void* stream) {
  const UirQkvFwdArgs& a = *p;
  UIR_REQUIRE(a.X && a.ln_g && a.ln_b && a.Wqkv && a.Wdw && a.Y);
  UIR_REQUIRE(!a.gram_slab || (a.C <= 32 && a.gram_stride >= (long)a.C * a.C + 2 * a.C));
  UIR_REQUIRE(a.B > 0 && a.H > 0 && a.Wd > 0);
  UIR_REQUIRE(a.ldx >= a.C && (a.ldx & 3) == 0 && uir_aligned16(a.X) && uir_aligned16(a.ln_g) && uir_aligned16(a.ln_b) &&
              uir_aligned16(a.Wqkv));
  UIR_REQUIRE((!a.Ypre || (a.ldp >= 3 * a.C && (a.ldp & 3) == 0 && uir_aligned16(a.Ypre))) && a.ldq >= 3 * a.C && (a.ldq & 3) == 0 && uir_aligned16(a.Y));
  UIR_REQUIRE((long)a.B * a.H * a.Wd * ((a.Ypre && a.ldp > a.ldq) ? a.ldp : a.ldq) < (1L << 30));
  hipStream_t st = static_cast<hipStream_t>(stream);
  switch (a.C) {
    case 16: return qkv_fwd_launch<16, 12, 16>(a, st);
    case 32: return qkv_fwd_launch<32, 12, 16>(a, st);
    case 64: return qkv_fwd_launch<64, 8, 16>(a, st);
    case 128: return qkv_fwd_launch<128, 4, 8>(a, st);
    default: return UIR_EINVAL;
  }
}

extern "C" int uir_attnv_fwd(const UirAttnvFwdArgs* p, void* stream) {
  const UirAttnvFwdArgs& a = *p;
  UIR_REQUIRE(a.V && a.WA && a.WB && a.Wdw && a.R && a.U && a.VF && a.Y);
  UIR_REQUIRE(a.B > 0 && a.H > 0 && a.Wd > 0);
  UIR_REQUIRE(a.ldv >= a.C && (a.ldv & 3) == 0 && uir_aligned16(a.V) && uir_aligned16(a.WA) && uir_aligned16(a.WB));
  UIR_REQUIRE(a.ldu >= a.C && a.ldf >= a.C && (a.ldu & 3) == 0 && (a.ldf & 3) == 0 && uir_aligned16(a.U) && uir_aligned16(a.VF));
  UIR_REQUIRE(a.ldr >= a.C && a.ldy >= a.C);
  UIR_REQUIRE((long)a.B * a.H * a.Wd * (a.ldu > a.ldf ? a.ldu : a.ldf) < (1L << 30));
  hipStream_t st = static_cast<hipStream_t>(stream);
  switch (a.C) {
    case 16: return attnv_fwd_launch<16, 12, 16>(a, st);
    case 32: return attnv_fwd_launch<32, 12, 16>(a, st);
    case 64: return attnv_fwd_launch<64, 8, 8>(a, st);
    case 128: return attnv_fwd_launch<128, 4, 8>(a, st);
    default: return UIR_EINVAL;
  }
}
