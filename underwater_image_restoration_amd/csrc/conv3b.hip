// uir_conv3x3_split*: UpSample1's dense 3 x 3 convolution (SpectralTransformer.py:191-198: Conv2d(C, 2C, 3, padding 1, no bias) at
// C = 128 / 64 / 32 on 32^2 / 64^2 / 128^2 pixels per image), its input gradient and its weight gradient as implicit GEMMs on
// split-bf16 MFMA.  9.66 GFLOP each; as nine-tap f32 MFMA launches (csrc/gemm.hip tap groups, csrc/tiled.hip) they ran at 42-70 TFLOP/s
// and the weight gradient (csrc/gram.hip nine-tap gathers) at 24.
//
// Forward / input gradient (uir_conv3b_kernel): a workgroup owns 4 or 8 image rows x 16 pixels x 32 .. 128 output channels.  Per 32-channel
// chunk of the input the HALO'd pixel tile is staged ONCE, split into three bf16 planes (x = h + m + l exactly), and all nine taps are
// walked from LDS: a tap is a row offset of the lane's ds_read_b128.  The weights are pre-split ONCE per optimizer step
// (uir_conv3x3_wsplit: planes [3][9][N][K], k-contiguous) and reach the MFMAs as 16-byte global loads of a lane's own fragment
// (L2-resident: 0.1 - 1.8 MB per layer), a step ahead of their use: no VALU work and no barrier per tap.  Six bf16 MFMAs per 16-deep k
// step reproduce the f32 product to 3 * 2^-24 (csrc/uir_common.h).
//
// Weight gradient (uir_conv3b_wgrad_kernel): the contraction runs over PIXELS, so both operands are needed pixel-major per lane while
// memory (and LDS, as staged) is channel-major: the 4 x 16-pixel tile of dY and the halo'd tile of X are read with the transposing
// ds_read_b64_tr_b16 (a lane supplies the address of one pixel row of a 4-pixel x 16-channel block and receives one channel of four pixels);
// a tap is again only a row offset.  A wave owns a 32 x 32 (n, k) block for all nine taps (144 accumulator registers) and writes it
// to its chunk's slab row in the native [n][k][tap] weight layout; uir_reduce_slab sums the chunks (deterministic).
#include "uir_common.h"
#include "../../include/uir_kernels.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

constexpr int TW = 16;                     // tile width (pixels)
constexpr int HWD = TW + 2;                // halo'd tile row
constexpr int SROW = 40;                   // forward kernel: bf16 per LDS pixel row: one 32-channel chunk (four 16-byte chunks) + 16 bytes of padding

__device__ __forceinline__ void split3(float x, uint32_t& h, uint32_t& m, uint32_t& l) {
  h = __float_as_uint(x) & 0xffff0000u;
  const float r1 = x - __uint_as_float(h);
  m = __float_as_uint(r1) & 0xffff0000u;
  l = __float_as_uint(r1 - __uint_as_float(m));          // at most 8 significant bits left: its top 16 bits hold it exactly
}
// four consecutive channels of one pixel -> 8 bytes at the same place of each of the three planes (PL bf16 apart)
template <int PL>
__device__ __forceinline__ void store_quad(unsigned short* d, const uint4& v, uint32_t mask) {
  const float x[4] = {__uint_as_float(v.x & mask), __uint_as_float(v.y & mask), __uint_as_float(v.z & mask), __uint_as_float(v.w & mask)};
  uint32_t h[4], m[4], l[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) split3(x[e], h[e], m[e], l[e]);
  *reinterpret_cast<uint2*>(d) = make_uint2(uir_pack_hi16(h[0], h[1]), uir_pack_hi16(h[2], h[3]));
  *reinterpret_cast<uint2*>(d + PL) = make_uint2(uir_pack_hi16(m[0], m[1]), uir_pack_hi16(m[2], m[3]));
  *reinterpret_cast<uint2*>(d + 2 * PL) = make_uint2(uir_pack_hi16(l[0], l[1]), uir_pack_hi16(l[2], l[3]));
}

// ------------------------------------------------------------------------------------------------ weights -> bf16 planes
__global__ __launch_bounds__(256) void uir_conv3b_wsplit_kernel(const float* __restrict__ W, int Cout, int Cin, unsigned short* __restrict__ pf,
                                                                unsigned short* __restrict__ pb) {
  // plane element (tap, n, k) of an [N][K] layer sits at ((tap * K / 16 + k / 16) * N + n) * 16 + k % 16: the 32 x 16 fragment a wave loads
  // for one MFMA step (lane = n, 16 bytes of one k half) is 1 KB CONTIGUOUS.  With plain [tap][n][k] rows a lane's 16 bytes were the only
  // part of its 128-byte line used per instruction and the 220 KB a workgroup walks do not live in L1: 8 x the bytes from L2 (ups_2: 27 TB/s)
  const long total = (long)Cout * Cin * 9, plane = total;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    if (pf) {                    // N = Cout, K = Cin:  i = ((tap * Cin / 16 + kb) * Cout + n) * 16 + kl
      const int kl = (int)(i & 15), n = (int)((i >> 4) % Cout);
      const long r = (i >> 4) / Cout;
      const int kb = (int)(r % (Cin / 16)), tap = (int)(r / (Cin / 16));
      uint32_t h, m, l;
      split3(W[((long)n * Cin + 16 * kb + kl) * 9 + tap], h, m, l);
      pf[i] = (unsigned short)(h >> 16); pf[plane + i] = (unsigned short)(m >> 16); pf[2 * plane + i] = (unsigned short)(l >> 16);
    }
    if (pb) {                    // N = Cin, K = Cout, tap' = 8 - tap:  i = ((tap' * Cout / 16 + kb) * Cin + n) * 16 + kl
      const int kl = (int)(i & 15), n = (int)((i >> 4) % Cin);
      const long r = (i >> 4) / Cin;
      const int kb = (int)(r % (Cout / 16)), tap = 8 - (int)(r / (Cout / 16));
      uint32_t h, m, l;
      split3(W[((long)(16 * kb + kl) * Cin + n) * 9 + tap], h, m, l);
      pb[i] = (unsigned short)(h >> 16); pb[plane + i] = (unsigned short)(m >> 16); pb[2 * plane + i] = (unsigned short)(l >> 16);
    }
  }
}

// ------------------------------------------------------------------------------------------------ forward / input gradient
// LDS image of the forward kernel: pixel rows of 80 bytes (64 used).  A ds_read_b128 is served in the lane groups {0-3, 12-15, 20-27},
// {4-11, 16-19, 28-31} (+32) over 64 banks: with MFMA row i of a 32-pixel tile holding pixel pix32(i), each group reads the same chunk of
// 16 CONSECUTIVE halo rows (one image row of the tile, any tap offset), i.e. the 16-byte slots (5 r + chunk) mod 16 - all distinct, conflict-free
// (pixel i in lane i on 64-byte XOR-swizzled rows was 2-way).  Without an XOR the address is affine in the tap: every (tap, k sub-block,
// plane) of a row tile is ONE base register + an immediate offset.
__device__ __forceinline__ int swz(int row, int chunk) { return row * SROW + (chunk << 3); }
__device__ __forceinline__ int pix32(int i) {      // lanes {0-3, 12-15, 20-27} -> pixels 0..15, lanes {4-11, 16-19, 28-31} -> 16..31
  return i < 4 ? i : i < 12 ? i + 12 : i < 16 ? i - 8 : i < 20 ? i + 8 : i < 28 ? i - 12 : i;
}

// the halo'd (TH + 2) x 18 pixel tile, one 32-channel chunk: (tid + 256 q) >> 3 = halo pixel, & 7 = channel quad
template <int TH>
struct HaloLoader {
  static constexpr int HP = (TH + 2) * HWD;
  static constexpr int NQ = (HP * 8 + 255) / 256;
  const float* base;
  uint32_t off[NQ];              // element offsets (the host checks that the tensor spans < 2^31 elements): one VGPR per load, scalar base
  uint32_t ok;                   // bit q: the pixel is inside the image (else zeros)
  __device__ __forceinline__ void init(const float* X, int ldx, int b, int y0, int x0, int H, int Wd, int tid) {
    ok = 0;
    base = X;
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      const int e = tid + 256 * q, hp = e >> 3;
      const int hy = hp / HWD, hx = hp - hy * HWD;
      const int gy = y0 + hy - 1, gx = x0 + hx - 1;
      const bool in = hp < HP && gy >= 0 && gy < H && gx >= 0 && gx < Wd;
      off[q] = (in ? (uint32_t)((b * H + gy) * Wd + gx) * (uint32_t)ldx : 0u) + 4u * (e & 7);      // outside: the tensor's first row (always readable)
      ok |= (in ? 1u : 0u) << q;
    }
  }
  __device__ __forceinline__ void load(uint4 (&v)[NQ], int k0) const {
#pragma unroll
    for (int q = 0; q < NQ; ++q) v[q] = *reinterpret_cast<const uint4*>(base + (off[q] + (uint32_t)k0));
  }
  template <int PL>
  __device__ __forceinline__ void store(unsigned short* s, const uint4 (&v)[NQ], int tid) const {
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      const int e = tid + 256 * q, hp = e >> 3;
      if ((q + 1) * 256 <= HP * 8 || hp < HP)
        store_quad<PL>(s + swz(hp, (e & 7) >> 1) + 4 * (e & 1), v[q], (ok >> q) & 1u ? 0xffffffffu : 0u);
    }
  }
};

template <int TMW, int TNW>
__global__ __launch_bounds__(256, 2) void uir_conv3b_kernel(const UirConv3bArgs a) {
  constexpr int TH = TMW / TW;
  constexpr int WR = TNW == 32 ? 4 : 2, WC = 4 / WR;        // waves as WR x WC
  constexpr int RT = TMW / (32 * WR), CT = TNW / (32 * WC);  // 32 x 32 tiles per wave
  constexpr int HP = (TH + 2) * HWD, PA = HP * SROW;
  constexpr int EW = 32 * CT + 4;                            // epilogue LDS row (floats)
  extern __shared__ float smem[];
  unsigned short* As = reinterpret_cast<unsigned short*>(smem);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, h2 = lane >> 5;
  const int wr = wave / WC, wc = wave % WC;
  const int ntn = a.N / TNW;
  const int tile = (int)blockIdx.x / ntn, n0 = ((int)blockIdx.x % ntn) * TNW;
  const int tpr = a.Wd / TW, tpi = tpr * (a.H / TH);
  const int b = tile / tpi, trem = tile - b * tpi;
  const int y0 = (trem / tpr) * TH, x0 = (trem % tpr) * TW;
  const int nkc = a.K / 32;

  HaloLoader<TH> la;
  la.init(a.X, a.ldx, b, y0, x0, a.H, a.Wd, tid);
  const unsigned short* ab[RT];                              // this lane's pixel at tap 0 (the halo row above-left), k sub-block h2
#pragma unroll
  for (int x = 0; x < RT; ++x) {
    const int m = 32 * (RT * wr + x) + pix32(li);
    ab[x] = As + swz((m >> 4) * HWD + (m & 15), h2);
  }
  const unsigned short* wb[CT];                              // this lane's weight row, k sub-block h2
#pragma unroll
  for (int x = 0; x < CT; ++x) wb[x] = a.Wp + (long)(n0 + 32 * (CT * wc + x) + li) * 16 + 8 * h2;
  const long stepst = (long)a.N * 16, tapst = (long)a.N * a.K, plst = 9 * tapst;      // planes [3][9][K / 16][N][16]

  f32x16 acc[RT][CT];
#pragma unroll
  for (int x = 0; x < RT; ++x)
#pragma unroll
    for (int y = 0; y < CT; ++y)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[x][y][r] = 0.f;

  uint4 va[HaloLoader<TH>::NQ];
  // weight fragments: a ring of PF steps, requested PF - 1 steps ahead.  A step is 6 RT CT MFMAs (192 .. 768 cycles) and an L2 hit takes
  // ~1 000: one step ahead the short steps waited for their weights (ups_1 on 64 x 128 tiles: 88 TFLOP/s)
  constexpr int PF = RT * CT >= 4 ? 2 : 3;
  uint4 fb[PF][CT][3], fa[RT][3];
  auto load_b = [&](uint4 (&dst)[CT][3], int kc, int s) {       // step s = (tap s / 2, k sub-block s % 2) of chunk kc
    const long off = (long)(s >> 1) * tapst + (2 * kc + (s & 1)) * stepst;
#pragma unroll
    for (int y = 0; y < CT; ++y)
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) dst[y][pl] = *reinterpret_cast<const uint4*>(wb[y] + pl * plst + off);
  };
  auto load_a = [&](int x, int s) {                             // compile-time offsets from one base register
    const int tap = s >> 1;
    const int off = swz((tap / 3) * HWD + tap % 3, 2 * (s & 1));
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) fa[x][pl] = *reinterpret_cast<const uint4*>(ab[x] + pl * PA + off);
  };
  la.load(va, 0);
#pragma unroll
  for (int i = 0; i < PF - 1; ++i) load_b(fb[i], 0, i);
  for (int kc = 0; kc < nkc; ++kc) {
    if (kc) __syncthreads();                       // every wave has read the previous chunk
    la.template store<PA>(As, va, tid);
    __syncthreads();
    const int kcn = kc + 1 < nkc ? kc + 1 : kc;           // past the end: clamped, loaded and never used
    la.load(va, kcn * 32);                                // next chunk's pixels in flight during the 18 steps
#pragma unroll
    for (int x = 0; x < RT; ++x) load_a(x, 0);
#pragma unroll
    for (int s = 0; s < 18; ++s) {
      __builtin_amdgcn_sched_barrier(0);
      if (s + PF - 1 < 18) load_b(fb[(s + PF - 1) % PF], kc, s + PF - 1);
      else load_b(fb[(s + PF - 1) % PF], kcn, s + PF - 1 - 18);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int x = 0; x < RT; ++x) {
        bf16x8 ca[3];
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) ca[pl] = __builtin_bit_cast(bf16x8, fa[x][pl]);
#pragma unroll
        for (int y = 0; y < CT; ++y) {
          bf16x8 cb[3];
#pragma unroll
          for (int pl = 0; pl < 3; ++pl) cb[pl] = __builtin_bit_cast(bf16x8, fb[s % PF][y][pl]);
          f32x16 c = acc[x][y];
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ca[1], cb[1], c, 0, 0, 0);     // small terms first
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ca[2], cb[0], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ca[0], cb[2], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ca[1], cb[0], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ca[0], cb[1], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ca[0], cb[0], c, 0, 0, 0);
          acc[x][y] = c;
        }
        // this row tile's pixels of step s + 1: re-read as soon as its own MFMAs have been issued, behind the other row tile's
        __builtin_amdgcn_sched_barrier(0);
        if (s < 17) load_a(x, s + 1);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
  __syncthreads();                                 // the pixel planes are dead: each wave's slice of them carries its epilogue

  // row-major epilogue through the wave's own LDS slice: float4 stores (csrc/tsgemm.hip)
  float* ws = smem + wave * (32 * EW);
  constexpr int LPR = 8 * CT, RPI = 64 / LPR, NI = 32 / RPI;
  const int c4 = 4 * (lane % LPR), rl = lane / LPR;
  const int col = n0 + 32 * CT * wc + c4;
#pragma unroll
  for (int x = 0; x < RT; ++x) {
#pragma unroll
    for (int y = 0; y < CT; ++y)
#pragma unroll
      for (int r = 0; r < 16; ++r) ws[uir_mfma_row(r, h2) * EW + 32 * y + li] = acc[x][y][r];
    __builtin_amdgcn_wave_barrier();
    const int mb = 32 * (RT * wr + x);               // first tile pixel of the band: two image rows of 16
    long rowp[NI];
    float4 rv[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int m = mb + pix32(rl + RPI * i);       // LDS row i of the band = MFMA row i = pixel pix32(i)
      rowp[i] = (long)(b * a.H + y0 + (m >> 4)) * a.Wd + x0 + (m & 15);
      rv[i] = a.R ? *reinterpret_cast<const float4*>(a.R + rowp[i] * a.ldr + col) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      float4 v = *reinterpret_cast<const float4*>(ws + (rl + RPI * i) * EW + c4);
      v.x += rv[i].x; v.y += rv[i].y; v.z += rv[i].z; v.w += rv[i].w;
      *reinterpret_cast<float4*>(a.Y + rowp[i] * a.ldy + col) = v;
    }
    __builtin_amdgcn_wave_barrier();
  }
}

template <int TMW, int TNW>
int conv3b_launch(const UirConv3bArgs& a, hipStream_t st) {
  constexpr int TH = TMW / TW, HP = (TH + 2) * HWD;
  constexpr int CT = TNW / (32 * (TNW == 32 ? 1 : 2));
  const size_t lds_op = (size_t)3 * HP * SROW * sizeof(unsigned short);
  const size_t lds_ep = (size_t)4 * 32 * (32 * CT + 4) * sizeof(float);
  const long wgs = (long)a.B * (a.H / TH) * (a.Wd / TW) * (a.N / TNW);
  hipLaunchKernelGGL((uir_conv3b_kernel<TMW, TNW>), dim3((unsigned)wgs), dim3(256), lds_op > lds_ep ? lds_op : lds_ep, st, a);
  return uir_launch_status();
}

// ------------------------------------------------------------------------------------------------ weight gradient
// LDS images: pixel rows of ROW bf16 (dY: 64 channels = 128 bytes = eight 16-byte chunks; X: 32 channels = 64 bytes).  A half-wave's
// transposed read takes 4 consecutive pixel rows x 64 bytes: on 128-byte rows the two 64-byte halves of rows r and r + 2 would share
// banks, so the halves swap where bit 1 of the row is set.
template <int ROW>
__device__ __forceinline__ int wsz(int row, int ch) {           // ch = channel (multiple of 4) -> bf16 offset
  const int chunk = ch >> 3;
  return row * ROW + (((ROW == 64) ? (chunk ^ (((row >> 1) & 1) << 2)) : chunk) << 3) + (ch & 7);
}

__device__ __forceinline__ s16x4 tr_read(const unsigned short* p) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p));
}
__device__ __forceinline__ bf16x8 pack8(s16x4 lo, s16x4 hi) {
  struct { s16x4 a, b; } t = {lo, hi};
  return __builtin_bit_cast(bf16x8, t);
}

// workgroup = 6 waves = 2 n tiles x 3 kernel rows over a 64 (n) x 32 (k) block of the gradient, persistent over 4 x 16 pixel tiles
constexpr int WG_THREADS = 384;
constexpr int WG_TH = 4, WG_HP = (WG_TH + 2) * HWD;             // halo'd 6 x 18
constexpr int WG_PD = 64 * 64, WG_PX = WG_HP * 32;              // bf16 per plane: dY tile [64 pixels][64 n], X tile [108 pixels][32 k]
constexpr int WG_NQD = (64 * 16 + WG_THREADS - 1) / WG_THREADS;    // float4 per thread: 3 (dY), 3 (X)
constexpr int WG_NQX = (WG_HP * 8 + WG_THREADS - 1) / WG_THREADS;

__global__ __launch_bounds__(WG_THREADS, 2) void uir_conv3b_wgrad_kernel(const UirConv3bWgradArgs a, const int ntiles) {
  extern __shared__ float smem[];
  unsigned short* Ds = reinterpret_cast<unsigned short*>(smem);
  unsigned short* Xs = Ds + 3 * WG_PD;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, h2 = lane >> 5;
  const int nt = wave & 1, trow = wave >> 1;                     // n tile, kernel row (dy = trow - 1)
  const int nkb = a.K / 32;
  const int n0 = 64 * ((int)blockIdx.y / nkb), k0 = 32 * ((int)blockIdx.y % nkb);
  const int tpr = a.Wd / TW, tpi = tpr * (a.H / WG_TH);
  // transposed-read lane roles: 16-lane group G, block row q, column quad p
  const int G = lane >> 4, q = (lane >> 2) & 3, p = lane & 3;
  const int acol = 32 * nt + 16 * (G & 1) + 4 * p, bcol = 16 * (G & 1) + 4 * p;

  f32x16 acc[3];
#pragma unroll
  for (int t = 0; t < 3; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  uint4 vd[WG_NQD], vx[WG_NQX];
  uint32_t okx = 0;
  auto load_tile = [&](int tile) {
    const int b = tile / tpi, trem = tile - b * tpi;
    const int y0 = (trem / tpr) * WG_TH, x0 = (trem % tpr) * TW;
#pragma unroll
    for (int i = 0; i < WG_NQD; ++i) {
      int e = tid + WG_THREADS * i;
      e = e < 64 * 16 ? e : 64 * 16 - 1;
      const int px = e >> 4, c4 = e & 15;
      const long row = (long)(b * a.H + y0 + (px >> 4)) * a.Wd + x0 + (px & 15);
      vd[i] = *reinterpret_cast<const uint4*>(a.dY + row * a.ldy + n0 + 4 * c4);
    }
    okx = 0;
#pragma unroll
    for (int i = 0; i < WG_NQX; ++i) {
      const int e = tid + WG_THREADS * i, hp = e >> 3, c4 = e & 7;
      const int hy = hp / HWD, hx = hp - hy * HWD;
      const int gy = y0 + hy - 1, gx = x0 + hx - 1;
      const bool in = hp < WG_HP && gy >= 0 && gy < a.H && gx >= 0 && gx < a.Wd;
      vx[i] = *reinterpret_cast<const uint4*>(a.X + (in ? ((long)(b * a.H + gy) * a.Wd + gx) * a.ldx : 0L) + k0 + 4 * c4);
      okx |= (in ? 1u : 0u) << i;
    }
  };
  auto store_tile = [&]() {
#pragma unroll
    for (int i = 0; i < WG_NQD; ++i) {
      const int e = tid + WG_THREADS * i, px = e >> 4, c4 = e & 15;
      if ((i + 1) * WG_THREADS <= 64 * 16 || e < 64 * 16) store_quad<WG_PD>(Ds + wsz<64>(px, 4 * c4), vd[i], 0xffffffffu);
    }
#pragma unroll
    for (int i = 0; i < WG_NQX; ++i) {
      const int e = tid + WG_THREADS * i, hp = e >> 3, c4 = e & 7;
      if ((i + 1) * WG_THREADS <= WG_HP * 8 || hp < WG_HP) store_quad<WG_PX>(Xs + wsz<32>(hp, 4 * c4), vx[i], (okx >> i) & 1u ? 0xffffffffu : 0u);
    }
  };

  int tile = blockIdx.x;
  load_tile(tile);                                        // nchunks <= ntiles: every workgroup has a first tile
  for (; tile < ntiles; tile += gridDim.x) {
    store_tile();
    __syncthreads();
    const int nxt = tile + (int)gridDim.x;
    load_tile(nxt < ntiles ? nxt : tile);               // next tile's pixels in flight during the MFMAs (clamped, unconditional)
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      // k slot (h2, e) of the 16-deep step = tile pixel (row s, column 8 h2 + e); this lane addresses block rows q and q + 4 of its group
      bf16x8 fa[3];
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) {
        const s16x4 lo = tr_read(Ds + pl * WG_PD + wsz<64>(16 * s + 8 * h2 + q, acol));
        const s16x4 hi = tr_read(Ds + pl * WG_PD + wsz<64>(16 * s + 8 * h2 + 4 + q, acol));
        fa[pl] = pack8(lo, hi);
      }
#pragma unroll
      for (int t = 0; t < 3; ++t) {
        const int hrow = (s + trow) * HWD + 8 * h2 + q + t;               // halo row of pixel (s + dy, 8 h2 + q + dx), dx = t - 1
        bf16x8 fbp[3];
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) {
          const s16x4 lo = tr_read(Xs + pl * WG_PX + wsz<32>(hrow, bcol));
          const s16x4 hi = tr_read(Xs + pl * WG_PX + wsz<32>(hrow + 4, bcol));
          fbp[pl] = pack8(lo, hi);
        }
        f32x16 c = acc[t];
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[1], fbp[1], c, 0, 0, 0);     // small terms first
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[2], fbp[0], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[0], fbp[2], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[1], fbp[0], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[0], fbp[1], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[0], fbp[0], c, 0, 0, 0);
        acc[t] = c;
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();                                    // every wave has read this tile
  }

  // the wave's 32 (n) x 32 (k) x 3 taps of the chunk's slab row, native [n][k][tap] order (a lane owns one k: 12 contiguous bytes per n)
  float* out = a.slab + (long)blockIdx.x * a.slab_stride;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int n = n0 + 32 * nt + uir_mfma_row(r, h2), k = k0 + li;
    float* o = out + ((long)n * a.K + k) * 9 + 3 * trow;
#pragma unroll
    for (int t = 0; t < 3; ++t) o[t] = acc[t][r];
  }
}

}  // namespace

extern "C" int uir_conv3x3_wsplit(const float* W, int Cout, int Cin, unsigned short* planes_fwd, unsigned short* planes_bwd, void* stream) {
  UIR_REQUIRE(W && Cout > 0 && Cin > 0 && (Cout & 15) == 0 && (Cin & 15) == 0 && (planes_fwd || planes_bwd));
  const long total = (long)Cout * Cin * 9;
  const int blocks = (int)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
  hipLaunchKernelGGL(uir_conv3b_wsplit_kernel, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream), W, Cout, Cin, planes_fwd, planes_bwd);
  return uir_launch_status();
}

extern "C" int uir_conv3x3_split(const UirConv3bArgs* p, void* stream) {
  const UirConv3bArgs& a = *p;
  UIR_REQUIRE(a.X && a.Wp && a.Y && a.B > 0 && a.H > 0 && a.Wd > 0);
  UIR_REQUIRE(a.N >= 32 && a.K >= 32 && (a.N & 31) == 0 && (a.K & 31) == 0 && (a.H & 3) == 0 && (a.Wd & 15) == 0);
  UIR_REQUIRE(a.ldx >= a.K && a.ldy >= a.N && (a.ldx & 3) == 0 && (a.ldy & 3) == 0 && uir_aligned16(a.X) && uir_aligned16(a.Y) && uir_aligned16(a.Wp));
  UIR_REQUIRE(!a.R || (a.ldr >= a.N && (a.ldr & 3) == 0 && uir_aligned16(a.R)));
  UIR_REQUIRE((long)a.B * a.H * a.Wd * (long)a.ldx < (1L << 31));      // 32-bit element offsets into X
  hipStream_t st = static_cast<hipStream_t>(stream);
  const bool h8 = (a.H & 7) == 0, h16 = (a.H & 15) == 0, n64 = (a.N & 63) == 0, n128 = (a.N & 127) == 0;
  auto wgs = [&](int tm, int tn) { return (long)a.B * (a.H / (tm / TW)) * (a.Wd / TW) * (a.N / tn); };
  int tile = a.tile;
  if (tile == 0) {
    // measured on the network's three layers and down3 (tools/bench_conv3b.py --tiles, profiles/r05): 128-pixel tiles always beat 64-pixel ones
    // (59-63 us against 102-111: half the MFMAs per weight fragment and a 1.7 x halo), 64 output columns tie with 128 and give twice the
    // workgroups; with 32 output columns the 256-pixel tile (one weight fragment against two pixel tiles per wave) wins from 256 workgroups on
    if (!n64) tile = (h16 && wgs(256, 32) >= 256) ? 256032 : 128032;
    else tile = h8 ? 128064 : 64064;
  }
  switch (tile) {
    case 256032: UIR_REQUIRE(h16); return conv3b_launch<256, 32>(a, st);
    case 128032: UIR_REQUIRE(h8); return conv3b_launch<128, 32>(a, st);
    case 128128: UIR_REQUIRE(h8 && n128); return conv3b_launch<128, 128>(a, st);
    case 64128: UIR_REQUIRE(n128); return conv3b_launch<64, 128>(a, st);
    case 128064: UIR_REQUIRE(h8 && n64); return conv3b_launch<128, 64>(a, st);
    case 64064: UIR_REQUIRE(n64); return conv3b_launch<64, 64>(a, st);
    default: return UIR_EINVAL;
  }
}

extern "C" int uir_conv3x3_split_wgrad(const UirConv3bWgradArgs* p, void* stream) {
  const UirConv3bWgradArgs& a = *p;
  UIR_REQUIRE(a.dY && a.X && a.slab && a.B > 0 && a.H > 0 && a.Wd > 0 && a.nchunks > 0);
  UIR_REQUIRE(a.N >= 64 && a.K >= 32 && (a.N & 63) == 0 && (a.K & 31) == 0 && (a.H & 3) == 0 && (a.Wd & 15) == 0);
  UIR_REQUIRE(a.ldy >= a.N && a.ldx >= a.K && (a.ldx & 3) == 0 && (a.ldy & 3) == 0 && uir_aligned16(a.X) && uir_aligned16(a.dY));
  UIR_REQUIRE(a.slab_stride >= (long)a.N * a.K * 9);
  const int ntiles = a.B * (a.H / WG_TH) * (a.Wd / TW);
  UIR_REQUIRE(a.nchunks <= ntiles);
  const size_t lds = (size_t)3 * (WG_PD + WG_PX) * sizeof(unsigned short);
  hipLaunchKernelGGL(uir_conv3b_wgrad_kernel, dim3(a.nchunks, (a.N / 64) * (a.K / 32)), dim3(WG_THREADS), lds, static_cast<hipStream_t>(stream), a, ntiles);
  return uir_launch_status();
}
