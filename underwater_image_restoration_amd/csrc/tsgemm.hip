// uir_ts_gemm: the "tall-skinny" channel contractions of the SpectralTransformer's deep levels (SpectralTransformer.py:81-83,86,90,
// 122,125 at C = 64 / 128: 16 384 - 65 536 pixel rows, 64 - 256 output channels, 64 - 704 input channels) on split-bf16 MFMA.
//
//     D[z][i][j] (+ R) = sum_k A[z][i][k] * B[z](j, k)         rows k-contiguous; B k-contiguous ([N][K]) or n-contiguous ([K][N])
//
// These launches are neither byte- nor f32-MFMA-bound on the panel kernel of csrc/gemm.hip (25 - 65 TFLOP/s on 128 - 512 workgroups,
// one dependent f32 MFMA chain per wave): they are short latency chains.  Here a workgroup owns a TMW x TNW tile (64 or 128 each, so
// that 16 384 rows x 128 columns are 256 workgroups), both operand chunks (32 k) are split into three bf16 planes when they are staged
// into LDS (csrc/tiled.hip's construction: 6 products reproduce the f32 product to 3 * 2^-24) and a chunk's 12 - 48 MFMAs run at 16x
// the f32 rate while the next chunk's loads are in flight.  Also covered, which csrc/tiled.hip's 128 x 128 kernel is not: per-sample
// weights (blockIdx.z: dqk / WA / WB / dv / dvf), the padded K groups of the GDFN tensors (dt2: pad columns of A are exact zeros, the
// weight row of a pad k is never real) and a residual.
#include "uir_common.h"
#include "../../include/uir_kernels.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
constexpr int SROW = 32;                   // bf16 per LDS row (64 bytes = four 16-byte chunks), XOR-swizzled as in csrc/tiled.hip
__device__ __forceinline__ int swz(int row, int chunk) { return (row ^ ((row >> 2) & 1)) * SROW + ((chunk ^ ((row >> 1) & 3)) << 3); }

__device__ __forceinline__ void split3(float x, uint32_t& h, uint32_t& m, uint32_t& l) {
  h = __float_as_uint(x) & 0xffff0000u;
  const float r1 = x - __uint_as_float(h);
  m = __float_as_uint(r1) & 0xffff0000u;
  l = __float_as_uint(r1 - __uint_as_float(m));
}
// four consecutive k of one row -> 8 bytes in each of the three planes (plane stride PL bf16)
template <int PL>
__device__ __forceinline__ void store_row4(unsigned short* s, int row, int k, const float (&x)[4]) {
  uint32_t h[4], m[4], l[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) split3(x[e], h[e], m[e], l[e]);
  unsigned short* d = s + swz(row, k >> 3) + (k & 7);
  *reinterpret_cast<uint2*>(d) = make_uint2(uir_pack_hi16(h[0], h[1]), uir_pack_hi16(h[2], h[3]));
  *reinterpret_cast<uint2*>(d + PL) = make_uint2(uir_pack_hi16(m[0], m[1]), uir_pack_hi16(m[2], m[3]));
  *reinterpret_cast<uint2*>(d + 2 * PL) = make_uint2(uir_pack_hi16(l[0], l[1]), uir_pack_hi16(l[2], l[3]));
}

struct TsArgs {
  const float* A; const float* B; float* D; const float* R;
  long sa_i, sb_j, sb_k, sd_i, ldr;
  long a_bs, b_bs, d_bs, r_bs;
  int MM, NN, KK;
  int kg_real, kg_pad;       // B rows exist for k % kg_pad < kg_real (kg_pad = 0: every k)
  int n_real;                // index-contiguous B: its rows hold n_real <= NN columns (GATE_BWD: the h real hidden channels of hp)
  const float* bias;         // [NN] or null (EPI 0)
  float slope;               // EPI 0: LeakyReLU slope applied to acc + bias (1 = none); EPI 1: the backward's slope
  const float* G; long ldg, g_bs; int g_off;   // EPI 1: forward LeakyReLU output; EPI 2: the GDFN tensor t2 (x1 half | x2 half at g_off); EPI 3: LN input
  const float* stats; const float* gamma;      // EPI 3: per-row (mean, rstd) of the LayerNorm, its weight
  float* slab; long slab_stride;               // EPI 3: per-workgroup partials [gridDim.x][2 NN] = dgamma | dbeta
};

// row of B's k axis that holds contraction index k (A's pad columns are exact zeros: a pad k may read any finite row)
__device__ __forceinline__ int kmap(int k, int real, int pad) {
  if (pad <= 0) return k;
  const int g = k / pad, r = k - g * pad;
  return g * real + (r < real ? r : real - 1);
}

// ROWS x 32 chunk of a k-contiguous operand: ROWS / 32 float4 per thread, raw loads from clamped addresses
template <int ROWS>
struct KcLoader {
  static constexpr int NQ = ROWS / 32;
  const float* p[NQ];
  __device__ __forceinline__ void init(const float* base, long s_idx, int idx0, int n_idx, int tid) {
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      int r = idx0 + ((tid + 256 * q) >> 3);
      r = r < n_idx ? r : n_idx - 1;
      p[q] = base + (long)r * s_idx + 4 * (tid & 7);
    }
  }
  __device__ __forceinline__ void load(uint4 (&v)[NQ], int k0, int ke) const {
    const int k = k0 + 4 * ((int)threadIdx.x & 7);
    const int kc = (k < ke ? k : ke - 4) - 4 * ((int)threadIdx.x & 7);
#pragma unroll
    for (int q = 0; q < NQ; ++q) v[q] = *reinterpret_cast<const uint4*>(p[q] + kc);
  }
  template <int PL>
  __device__ __forceinline__ void store(unsigned short* s, const uint4 (&v)[NQ], int tid, int k0, int ke) const {
    const uint32_t mk = k0 + 4 * (tid & 7) < ke ? 0xffffffffu : 0u;
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      const float x[4] = {__uint_as_float(v[q].x & mk), __uint_as_float(v[q].y & mk), __uint_as_float(v[q].z & mk), __uint_as_float(v[q].w & mk)};
      store_row4<PL>(s, (tid + 256 * q) >> 3, 4 * (tid & 7), x);
    }
  }
};

// ROWS x 32 chunk of an index-contiguous operand (element (j, k) at base[j + kmap(k) * s_k]): the ROWS / 4 index quads x 8 blocks of 4 k
// are dealt to the first ROWS * 2 threads, each loading 4 float4 (4 consecutive k of one index quad) and transposing in registers
template <int ROWS>
struct IcLoader {
  const float* p;
  int row0, kb, shift;
  bool on;
  // n_idx: columns that exist in memory.  A quad that reaches past them is loaded from the last whole quad and its elements are
  // shifted into place when they are stored (columns past n_idx become zeros): no read past the operand, no wrong column
  __device__ __forceinline__ void init(const float* base, int idx0, int n_idx, int tid) {
    on = tid < ROWS * 2;
    const int quad = tid >> 3;
    kb = 4 * (tid & 7);
    row0 = 4 * (on ? quad : 0);
    const int i = idx0 + row0;
    const int ic = i + 4 <= n_idx ? i : n_idx - 4;
    shift = i - ic;                                        // 0 .. 3 inside the operand, >= 4: the whole quad is past it
    p = base + ic;
  }
  __device__ __forceinline__ void load(uint4 (&v)[4], int k0, int ke, long s_k, int real, int pad) const {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int k = k0 + kb + q;
      v[q] = *reinterpret_cast<const uint4*>(p + (long)kmap(k < ke ? k : ke - 1, real, pad) * s_k);
    }
  }
  template <int PL>
  __device__ __forceinline__ void store(unsigned short* s, const uint4 (&v)[4], int k0, int ke) const {
    if (!on) return;
    uint32_t mk[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) mk[q] = k0 + kb + q < ke ? 0xffffffffu : 0u;
    float r0[4], r1[4], r2[4], r3[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const uint32_t c0 = v[q].x & mk[q], c1 = v[q].y & mk[q], c2 = v[q].z & mk[q], c3 = v[q].w & mk[q];
      // element e of the quad is loaded element e + shift
      r0[q] = __uint_as_float(shift == 0 ? c0 : shift == 1 ? c1 : shift == 2 ? c2 : shift == 3 ? c3 : 0u);
      r1[q] = __uint_as_float(shift == 0 ? c1 : shift == 1 ? c2 : shift == 2 ? c3 : 0u);
      r2[q] = __uint_as_float(shift == 0 ? c2 : shift == 1 ? c3 : 0u);
      r3[q] = __uint_as_float(shift == 0 ? c3 : 0u);
    }
    store_row4<PL>(s, row0, kb, r0);
    store_row4<PL>(s, row0 + 1, kb, r1);
    store_row4<PL>(s, row0 + 2, kb, r2);
    store_row4<PL>(s, row0 + 3, kb, r3);
  }
};

// EPI 0: D = leaky(acc + bias) + R;  1: LeakyReLU backward (D = G > 0 ? acc : slope acc);  2: GDFN gate backward (two outputs);
// 3: LayerNorm backward on the accumulators (the tile holds whole rows: NN == TNW) - D = R + LN'(acc) with the LayerNorm's input G, its
//    saved statistics and weight, and the workgroup's dgamma | dbeta partial sums to one slab row (uir_layernorm_bwd's arithmetic: the
//    deep levels' dt1 / dt2 contractions were followed by that kernel, 20 launches and 0.3 ms of the chain per step)
template <bool BKC, int TMW, int TNW, int EPI>
__global__ __launch_bounds__(256, 3) void uir_ts_split_kernel(const TsArgs t) {
  constexpr int RT = TMW / 64, CT = TNW / 64;              // 32-row / 32-column tiles per wave (2 x 2 waves)
  constexpr int PA = TMW * SROW, PB = TNW * SROW;          // bf16 per plane
  constexpr int EW = 32 * CT + 4;                          // epilogue LDS row (floats)
  extern __shared__ float smem[];
  unsigned short* As = reinterpret_cast<unsigned short*>(smem);
  unsigned short* Bs = As + 3 * PA;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, h2 = lane >> 5;
  const int wr = wave >> 1, wc = wave & 1;
  const int tiles_j = (t.NN + TNW - 1) / TNW;
  const int i0 = ((int)blockIdx.x / tiles_j) * TMW, j0 = ((int)blockIdx.x % tiles_j) * TNW;
  const int z = blockIdx.z;
  const float* A = t.A + (long)z * t.a_bs;
  const float* B = t.B + (long)z * t.b_bs;
  const int ke = t.KK;

  KcLoader<TMW> la;
  la.init(A, t.sa_i, i0, t.MM, tid);
  KcLoader<TNW> lbk;
  IcLoader<TNW> lbi;
  if constexpr (BKC) lbk.init(B, t.sb_j, j0, t.NN, tid);
  else lbi.init(B, j0, t.n_real, tid);

  f32x16 acc[RT][CT];
#pragma unroll
  for (int a = 0; a < RT; ++a)
#pragma unroll
    for (int b = 0; b < CT; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  uint4 va[TMW / 32], vb[BKC ? TNW / 32 : 4];
  auto load_b = [&](int k0) {
    if constexpr (BKC) lbk.load(vb, k0, ke);
    else lbi.load(vb, k0, ke, t.sb_k, t.kg_real, t.kg_pad);
  };
  auto store_b = [&](int k0) {
    if constexpr (BKC) lbk.template store<PB>(Bs, vb, tid, k0, ke);
    else lbi.template store<PB>(Bs, vb, k0, ke);
  };
  la.load(va, 0, ke);
  load_b(0);
  la.template store<PA>(As, va, tid, 0, ke);
  store_b(0);
  __syncthreads();
  for (int k0 = 0; k0 < ke; k0 += 32) {
    const bool more = k0 + 32 < ke;
    la.load(va, k0 + 32, ke);                       // unconditional, clamped: never under a branch (csrc/tiled.hip)
    load_b(k0 + 32);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int kq = 0; kq < 2; ++kq) {
      bf16x8 fa[RT][3], fb[CT][3];
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) {
#pragma unroll
        for (int x = 0; x < RT; ++x)
          fa[x][pl] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(As + pl * PA + swz(32 * RT * wr + 32 * x + li, 2 * kq + h2)));
#pragma unroll
        for (int x = 0; x < CT; ++x)
          fb[x][pl] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(Bs + pl * PB + swz(32 * CT * wc + 32 * x + li, 2 * kq + h2)));
      }
#pragma unroll
      for (int a = 0; a < RT; ++a)
#pragma unroll
        for (int b = 0; b < CT; ++b) {
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
      la.template store<PA>(As, va, tid, k0 + 32, ke);
      store_b(k0 + 32);
    }
    __syncthreads();
  }

  if constexpr (EPI == 3) {
    // LayerNorm backward.  A row's NN columns live in the two waves wc = 0 / 1 of a row pair: each wave reduces its half over the LPR lanes
    // of the row, the halves meet in LDS (one workgroup barrier per 32-row band; every wave walks the same bands)
    float* ws = smem + wave * (32 * EW);
    float* xr = smem + 4 * 32 * EW;                            // [band parity][wave][32 rows][2]
    float* cred = xr + 2 * 4 * 32 * 2;                         // [wave][64 lanes][8]: column partials at the end
    constexpr int LPR = 8 * CT, RPI = 64 / LPR, NI = 32 / RPI;
    const int c4 = 4 * (lane % LPR), rl = lane / LPR;
    const int col = 32 * CT * wc + c4;                         // j0 == 0: the tile is the whole row
    const float invc = 1.0f / (float)t.NN;
    const float4 gam = *reinterpret_cast<const float4*>(t.gamma + col);
    float dg[4] = {0.f, 0.f, 0.f, 0.f}, db[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int a = 0; a < RT; ++a) {
#pragma unroll
      for (int b = 0; b < CT; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) ws[uir_mfma_row(r, h2) * EW + 32 * b + li] = acc[a][b][r];
      __builtin_amdgcn_wave_barrier();
      const int rb = i0 + 32 * RT * wr + 32 * a;
      float4 dt[NI], xh[NI], rv[NI];
      float rs[NI], s1[NI], s2[NI];
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        const int rr = rl + RPI * i;
        const int row = min(rb + rr, t.MM - 1);
        const float4 x = *reinterpret_cast<const float4*>(t.G + (long)row * t.ldg + col);
        const float2 st = *reinterpret_cast<const float2*>(t.stats + 2L * row);
        rv[i] = t.R ? *reinterpret_cast<const float4*>(t.R + (long)row * t.ldr + col) : make_float4(0.f, 0.f, 0.f, 0.f);
        dt[i] = *reinterpret_cast<const float4*>(ws + rr * EW + c4);
        rs[i] = st.y;
        xh[i] = make_float4((x.x - st.x) * st.y, (x.y - st.x) * st.y, (x.z - st.x) * st.y, (x.w - st.x) * st.y);
        const float4 gd = make_float4(dt[i].x * gam.x, dt[i].y * gam.y, dt[i].z * gam.z, dt[i].w * gam.w);
        float p1 = (gd.x + gd.y) + (gd.z + gd.w);
        float p2 = (gd.x * xh[i].x + gd.y * xh[i].y) + (gd.z * xh[i].z + gd.w * xh[i].w);
#pragma unroll
        for (int o = LPR / 2; o > 0; o >>= 1) { p1 += __shfl_xor(p1, o); p2 += __shfl_xor(p2, o); }
        s1[i] = p1; s2[i] = p2;
        if ((lane % LPR) == 0) { xr[(((a & 1) * 4 + wave) * 32 + rr) * 2] = p1; xr[(((a & 1) * 4 + wave) * 32 + rr) * 2 + 1] = p2; }
      }
      __syncthreads();
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        const int rr = rl + RPI * i, row = rb + rr;
        const float* o = xr + (((a & 1) * 4 + (wave ^ 1)) * 32 + rr) * 2;       // the other half of the row: wave with the other wc
        // fixed order (wc = 0 half first) so that both waves form the same sums
        const float t1 = (wc == 0 ? s1[i] + o[0] : o[0] + s1[i]) * invc, t2 = (wc == 0 ? s2[i] + o[1] : o[1] + s2[i]) * invc;
        const float4 gd = make_float4(dt[i].x * gam.x, dt[i].y * gam.y, dt[i].z * gam.z, dt[i].w * gam.w);
        float4 dx = make_float4(rs[i] * (gd.x - t1 - xh[i].x * t2), rs[i] * (gd.y - t1 - xh[i].y * t2),
                                rs[i] * (gd.z - t1 - xh[i].z * t2), rs[i] * (gd.w - t1 - xh[i].w * t2));
        dx.x += rv[i].x; dx.y += rv[i].y; dx.z += rv[i].z; dx.w += rv[i].w;
        if (row < t.MM) {
          *reinterpret_cast<float4*>(t.D + (long)row * t.sd_i + col) = dx;
          dg[0] += dt[i].x * xh[i].x; dg[1] += dt[i].y * xh[i].y; dg[2] += dt[i].z * xh[i].z; dg[3] += dt[i].w * xh[i].w;
          db[0] += dt[i].x; db[1] += dt[i].y; db[2] += dt[i].z; db[3] += dt[i].w;
        }
      }
      __builtin_amdgcn_wave_barrier();
    }
    // dgamma | dbeta of the workgroup's rows: lanes with the same column quad (RPI per wave, two waves wr) in a fixed order
    float* cp = cred + (wave * 64 + lane) * 8;
#pragma unroll
    for (int k = 0; k < 4; ++k) { cp[k] = dg[k]; cp[4 + k] = db[k]; }
    __syncthreads();
    for (int e = tid; e < 2 * t.NN; e += 256) {
      const int which = e / t.NN, c = e - which * t.NN;
      const int wcc = c / (32 * CT), q = (c - 32 * CT * wcc) >> 2, k = c & 3;
      float sum = 0.f;
#pragma unroll
      for (int wrr = 0; wrr < 2; ++wrr)
        for (int r2 = 0; r2 < RPI; ++r2) sum += cred[((2 * wrr + wcc) * 64 + r2 * LPR + q) * 8 + 4 * which + k];
      t.slab[(long)blockIdx.x * t.slab_stride + e] = sum;
    }
    return;
  }
  // row-major epilogue through the wave's own LDS slice (the operand planes are dead behind the loop's last barrier): float4 stores
  float* D = t.D + (long)z * t.d_bs;
  const float* R = t.R ? t.R + (long)z * t.r_bs : nullptr;
  float* ws = smem + wave * (32 * EW);
  constexpr int LPR = 8 * CT, RPI = 64 / LPR, NI = 32 / RPI;
  const int c4 = 4 * (lane % LPR), rl = lane / LPR;
  const int col = j0 + 32 * CT * wc + c4;
  const bool cok = col < t.NN;                               // NN is a multiple of 4
#pragma unroll
  for (int a = 0; a < RT; ++a) {
#pragma unroll
    for (int b = 0; b < CT; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) ws[uir_mfma_row(r, h2) * EW + 32 * b + li] = acc[a][b][r];
    __builtin_amdgcn_wave_barrier();
    const int rb = i0 + 32 * RT * wr + 32 * a;
    // the band's extra operands (residual / forward output / gate tensor) in flight together, before the first store (they may alias D)
    float4 rv[NI], gv[EPI == 2 ? NI : 1];
    const float* G = (EPI != 0) ? t.G + (long)z * t.g_bs : nullptr;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int row = rb + rl + RPI * i;
      const bool ok = cok && row < t.MM;
      if (EPI == 0) rv[i] = (R && ok) ? *reinterpret_cast<const float4*>(R + (long)row * t.ldr + col) : make_float4(0.f, 0.f, 0.f, 0.f);
      else rv[i] = ok ? *reinterpret_cast<const float4*>(G + (long)row * t.ldg + col) : make_float4(0.f, 0.f, 0.f, 0.f);
      if (EPI == 2) gv[i] = ok ? *reinterpret_cast<const float4*>(G + (long)row * t.ldg + t.g_off + col) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    float4 bj = make_float4(0.f, 0.f, 0.f, 0.f);
    if (EPI == 0 && t.bias && cok) bj = *reinterpret_cast<const float4*>(t.bias + col);
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int rr = rl + RPI * i, row = rb + rr;
      float4 v = *reinterpret_cast<const float4*>(ws + rr * EW + c4);
      if (cok && row < t.MM) {
        if (EPI == 0) {
          v.x += bj.x; v.y += bj.y; v.z += bj.z; v.w += bj.w;
          v.x = v.x > 0.f ? v.x : v.x * t.slope; v.y = v.y > 0.f ? v.y : v.y * t.slope;
          v.z = v.z > 0.f ? v.z : v.z * t.slope; v.w = v.w > 0.f ? v.w : v.w * t.slope;
          v.x += rv[i].x; v.y += rv[i].y; v.z += rv[i].z; v.w += rv[i].w;
          *reinterpret_cast<float4*>(D + (long)row * t.sd_i + col) = v;
        } else if (EPI == 1) {
          v.x = rv[i].x > 0.f ? v.x : v.x * t.slope; v.y = rv[i].y > 0.f ? v.y : v.y * t.slope;
          v.z = rv[i].z > 0.f ? v.z : v.z * t.slope; v.w = rv[i].w > 0.f ? v.w : v.w * t.slope;
          *reinterpret_cast<float4*>(D + (long)row * t.sd_i + col) = v;
        } else {
          // hidden channels col .. col + 3 of the x1 half; pad channels (>= n_real) give exact zeros in both halves
          const float a4[4] = {v.x, v.y, v.z, v.w}, g1[4] = {rv[i].x, rv[i].y, rv[i].z, rv[i].w}, g2[4] = {gv[i].x, gv[i].y, gv[i].z, gv[i].w};
          float o1[4], o2[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            float ge, dge;
            uir_gelu_pair(g1[e], ge, dge);
            const float av = col + e < t.n_real ? a4[e] : 0.f;
            o1[e] = av * g2[e] * dge;
            o2[e] = av * ge;
          }
          *reinterpret_cast<float4*>(D + (long)row * t.sd_i + col) = make_float4(o1[0], o1[1], o1[2], o1[3]);
          *reinterpret_cast<float4*>(D + (long)row * t.sd_i + t.g_off + col) = make_float4(o2[0], o2[1], o2[2], o2[3]);
        }
      }
    }
    __builtin_amdgcn_wave_barrier();
  }
}

template <bool BKC, int TMW, int TNW, int EPI>
int ts_launch(const TsArgs& t, int batches, hipStream_t st) {
  const dim3 grid(((t.MM + TMW - 1) / TMW) * ((t.NN + TNW - 1) / TNW), 1, batches);
  const size_t lds_op = (size_t)3 * (TMW + TNW) * SROW * sizeof(unsigned short);
  size_t lds_ep = (size_t)4 * 32 * (32 * (TNW / 64) + 4) * sizeof(float);
  if (EPI == 3) lds_ep += (size_t)(2 * 4 * 32 * 2 + 4 * 64 * 8) * sizeof(float);       // row-sum exchange + column partials
  hipLaunchKernelGGL((uir_ts_split_kernel<BKC, TMW, TNW, EPI>), grid, dim3(256), lds_op > lds_ep ? lds_op : lds_ep, st, t);
  return uir_launch_status();
}

}  // namespace

// -1: not this kernel's shape (the caller continues on the panel kernel).  Plain epilogue only: no bias / activation / prologue.
int uir_ts_gemm(const UirGemmArgs& a, void* stream) {
  if (!uir_fused_split() || uir_mixed_mode() != 0) return -1;      // UIR_MIXED: the 16-bit kernel of csrc/tiled.hip takes these calls
  if (a.taps != 1 || a.pro != UIR_PRO_NONE || a.out_scale || a.pre_out || a.il16) return -1;
  const int epi = a.epi == UIR_EPI_PLAIN ? 0 : a.epi == UIR_EPI_LEAKY_BWD ? 1 : a.epi == UIR_EPI_GATE_BWD ? 2 : a.epi == UIR_EPI_LN_BWD ? 3 : -1;
  if (epi < 0 || (a.act != UIR_ACT_NONE && (a.act != UIR_ACT_LEAKY || epi != 0)) || (epi != 0 && (a.bias || (a.R && epi != 3) || !a.G || a.batches != 1))) return -1;
  // LayerNorm backward: the tile is the whole row
  if (epi == 3 && ((a.N != 64 && a.N != 128) || a.n_grp_pad > 0 || !a.stats || !a.ln_g || !uir_aligned16(a.ln_g) || !a.dw_slab || a.dw_stride < 2 * a.N ||
                   (a.ldg & 3) || a.ldg < a.N || !uir_aligned16(a.G))) return -1;
  // column groups only as the gate backward has them: one group, N = hp columns of which the first h are real
  if (a.n_grp_pad > 0 && !(epi == 2 && a.n_grp_pad == a.N && a.w_sn == 1)) return -1;
  if (epi == 2 && ((a.g_off & 3) || (a.ldg & 3) || !uir_aligned16(a.G) || a.ldg < a.g_off + a.N || a.ldy < a.g_off + a.N)) return -1;
  if (epi == 1 && ((a.ldg & 3) || !uir_aligned16(a.G) || a.ldg < a.N)) return -1;
  if (a.bias && !uir_aligned16(a.bias)) return -1;
  const long rows = (long)a.M * a.batches;
  // the deep levels: few rows (an under-filled chip on the panel kernel), contractions long enough to pay for the operand splits
  // ... and, on any number of rows, long contractions into fewer than 128 columns (the module networks' Linear input gradients, 512 -> 64 on 524 288
  // tokens): csrc/tiled.hip takes N >= 128 only and the panel kernel runs them at ~60 TFLOP/s of f32 MFMA against 2.2 TB/s of operands
  const bool narrow_long = a.N < 128 && a.K >= 256 && a.batches == 1 && rows < (1L << 22);
  if (rows < 4096 || (rows > 65536 && !narrow_long) || a.M < 256 || a.K < 64 || a.N < 64 || (a.N & 3) || (a.K & 3)) return -1;
  const bool bkc = a.w_sk == 1 && (a.w_sn & 3) == 0, bic = a.w_sn == 1 && (a.w_sk & 3) == 0;
  if (!bkc && !bic) return -1;
  if (a.k_grp_pad > 0 && (!bic || (a.K % a.k_grp_pad) != 0)) return -1;
  if ((a.ldx & 3) || (a.ldy & 3) || !uir_aligned16(a.X) || !uir_aligned16(a.W) || !uir_aligned16(a.Y) || (a.x_bs & 3) || (a.y_bs & 3) || (a.w_bs & 3)) return -1;
  if (a.R && ((a.ldr & 3) || !uir_aligned16(a.R) || (a.r_bs & 3))) return -1;
  TsArgs t{};
  t.A = a.X; t.B = a.W; t.D = a.Y; t.R = a.R;
  t.sa_i = a.ldx; t.sb_j = a.w_sn; t.sb_k = a.w_sk; t.sd_i = a.ldy; t.ldr = a.ldr;
  t.a_bs = a.x_bs; t.b_bs = a.w_bs; t.d_bs = a.y_bs; t.r_bs = a.r_bs;
  t.MM = a.M; t.NN = a.N; t.KK = a.K;
  t.kg_real = a.k_grp_real; t.kg_pad = a.k_grp_pad;
  t.n_real = a.n_grp_pad > 0 ? a.n_grp_real : a.N;
  if (t.n_real < 4) return -1;
  t.bias = a.bias;
  t.slope = (epi == 0 && a.act != UIR_ACT_LEAKY) ? 1.0f : a.slope;
  t.G = a.G; t.ldg = a.ldg; t.g_bs = a.g_bs; t.g_off = a.g_off;
  t.stats = a.stats; t.gamma = a.ln_g; t.slab = a.dw_slab; t.slab_stride = a.dw_stride;
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (epi == 3) {
    // one column tile = the row; 128-row tiles while they still give two workgroups per CU.  Slab rows = row tiles (<= M / 64)
    const bool m128 = (long)((a.M + 127) / 128) >= 512;
    const bool bic3 = a.w_sn == 1 && (a.w_sk & 3) == 0;
    if (!bic3 && !(a.w_sk == 1 && (a.w_sn & 3) == 0)) return -1;
    if (a.N == 64) {
      if (bic3) return m128 ? ts_launch<false, 128, 64, 3>(t, 1, st) : ts_launch<false, 64, 64, 3>(t, 1, st);
      return m128 ? ts_launch<true, 128, 64, 3>(t, 1, st) : ts_launch<true, 64, 64, 3>(t, 1, st);
    }
    if (bic3) return m128 ? ts_launch<false, 128, 128, 3>(t, 1, st) : ts_launch<false, 64, 128, 3>(t, 1, st);
    return m128 ? ts_launch<true, 128, 128, 3>(t, 1, st) : ts_launch<true, 64, 128, 3>(t, 1, st);
  }
  // tile: the largest of 128 x 128, 64 x 128 / 128 x 64, 64 x 64 that still gives the chip two workgroups per CU (512): these launches
  // are latency chains of K / 32 chunks, and a second workgroup on the CU runs under the first one's barriers and load waits
  // (level 3's dt1, 16 384 x 384 -> 128: 256 workgroups of 64 x 128 lost to the f32 kernel of csrc/tiled.hip, 512 of 64 x 64 do not).
  // 64-wide column tiles also where a 128-wide one would be at least half empty.
  auto wgs = [&](int tm, int tn) { return (long)((a.M + tm - 1) / tm) * ((a.N + tn - 1) / tn) * a.batches; };
  const bool nfit128 = a.N >= 128 && !((a.N % 128) != 0 && (a.N % 128) <= 64);
  bool m64, n64;
  if (nfit128 && wgs(128, 128) >= 512) { m64 = false; n64 = false; }
  else if (nfit128 && wgs(64, 128) >= 512) { m64 = true; n64 = false; }
  else if (wgs(128, 64) >= 512) { m64 = false; n64 = true; }
  else { m64 = true; n64 = true; }
#define UIR_TS(BK, EP) do { \
    if (m64 && n64) return ts_launch<BK, 64, 64, EP>(t, a.batches, st); \
    if (m64) return ts_launch<BK, 64, 128, EP>(t, a.batches, st); \
    if (n64) return ts_launch<BK, 128, 64, EP>(t, a.batches, st); \
    return ts_launch<BK, 128, 128, EP>(t, a.batches, st); } while (0)
  if (epi == 2) { if (bkc) return -1; UIR_TS(false, 2); }        // the gate backward contracts against W_out^T: n-contiguous
  if (epi == 1) { if (bkc) UIR_TS(true, 1); UIR_TS(false, 1); }
  if (bkc) UIR_TS(true, 0);
  UIR_TS(false, 0);
#undef UIR_TS
}
