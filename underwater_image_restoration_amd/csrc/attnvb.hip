// uir_attnv_bwd: the backward of MDTA's pixel-sized tail (uir_attnv_fwd:  u = v . WA[b]^T,  vf = dw3x3(u),  x1 = vf . WB[b]^T + x;
// SpectralTransformer.py:100,107,110,113 under autograd) in ONE pass over the pixels, C = 16 / 32 (below) and C = 64 (uir_attnv_bwd64_kernel):
//
//   dvf = dx1 . WB[b]        du = dw3x3^T(dvf)        dv = du . WA[b]                              the value path's input gradient
//   dWB[b][i][j] = sum_p dx1[p][i] vf[p][j]       dWA[b][j][o] = sum_p du[p][j] v[p][o]            per-sample C x C gradients (uir_attn_bwd reads them)
//   dWdw[j][t]   = sum_p dvf[p - off(t)][j] u[p][j]                                               depthwise weight gradient
//
// It replaces three launches - the dvf GEMM with its fused Gram, uir_dwconv_bwd_fused, the dv GEMM with its fused Gram (9 tensors of C floats
// per pixel: each intermediate written once and read once) - by one that reads dx1 (tile + ring), vf, u, v and writes dv (5.4 tensors).
// uir_attnv_fwd's skeleton mirrored: a workgroup walks 8 x 16 pixel tiles of ONE image (persistent: its weight-gradient partial sums stay in
// registers until the end, one slab row per workgroup);
//   (a) dvf on tile + ring   split-bf16 MFMA from the dx1 rows held in registers                  -> LDS
//   (b) du = transposed depthwise 3x3 from LDS (rotating three-row window, packed f32 over the channel pair); the same window elements times
//       the centre u value are the nine depthwise weight-gradient terms                           -> LDS
//   (c) dv = du . WA         split-bf16 MFMA, stored from the accumulators
//   (d) the two per-sample Grams on the f32 MFMA, a quarter of the tile's pixels per wave (operands read from LDS one value per lane:
//       lane = channel, k slot = pixel, the layout of uir_qkv_fwd's logit Gram)
// Pixels outside the image get dvf := 0 (the zero padding of the forward convolution).  H % 8 == 0, W % 16 == 0 (C = 64: W % 8 == 0).
#include "uir_common.h"
#include "../../include/uir_kernels.h"

namespace {

__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
__device__ __forceinline__ float4 keep4(const float4& t, bool ok) {
  const unsigned m = ok ? 0xffffffffu : 0u;
  return make_float4(__uint_as_float(__float_as_uint(t.x) & m), __uint_as_float(__float_as_uint(t.y) & m),
                     __uint_as_float(__float_as_uint(t.z) & m), __uint_as_float(__float_as_uint(t.w) & m));
}

template <int C>
struct AvbTile {
  static constexpr int TH = 8, TW = 16, RW = TW + 2, RH = TH + 2;
  static constexpr int NPX1 = RW * RH;              // 180 tile + ring pixels
  static constexpr int NRT1 = (NPX1 + 31) / 32;     // 6 row tiles of 32
  static constexpr int RTW = (NRT1 + 3) / 4;        // 2 per wave
  static constexpr int NI = TH * TW;                // 128 interior pixels = 4 row tiles, one per wave
  static constexpr int S1 = 40, SG = 36;            // LDS row strides (floats): dvf ring rows / interior rows
  static constexpr int KJ = C / 8;
  static constexpr int NST = NI * 8 / 256;          // float4 per thread of an interior [128][32] tile (4)
  static constexpr int OFF_G = NRT1 * 32 * S1;      // du
  static constexpr int OFF_R = OFF_G + NI * SG;     // u, then v, then vf
  static constexpr int LDS_FLOATS = OFF_R + NI * SG;
  static_assert(C == 16 || C == 32, "channel count");
};

template <int C>
__global__ __launch_bounds__(256, 2) void uir_attnv_bwd_kernel(const UirAttnvBwdArgs a, const int wps, const int tiles_x, const int tiles_img) {
  using T = AvbTile<C>;
  constexpr int TW = T::TW;
  extern __shared__ float smem[];
  float* T1s = smem;                    // dvf on tile + ring [192][S1]; between tiles: the waves' Gram partials
  float* Gs = smem + T::OFF_G;          // du [128][SG]; then dx1 interior
  float* Rs = smem + T::OFF_R;          // u, v, vf in turn [128][SG]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, h2 = lane >> 5;
  const int b = (int)blockIdx.x / wps, w = (int)blockIdx.x % wps;
  const long img = (long)b * a.H * a.Wd;
  const float* WA = a.WA + (long)b * C * C;
  const float* WB = a.WB + (long)b * C * C;
  const bool cok = li < C;              // MFMA column / Gram row of this lane exists (C = 16: half the tile)

  // ---- per-workgroup operands: B fragments of (a) and (c), k slot t of split step s <-> k = 16 s + 8 (t / 4) + 4 h2 + t % 4 (uir_attnv_fwd's map) ----
  UirBf3 wpa[C / 16], wpc[C / 16];
#pragma unroll
  for (int s = 0; s < C / 16; ++s) {
    float xa[8], xc[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      const int k = 16 * s + 8 * (t >> 2) + 4 * h2 + (t & 3);
      xa[t] = cok ? WB[k * C + li] : 0.f;            // (a): dvf[p][j = li] = sum_i dx1[p][i] WB[i][j]
      xc[t] = cok ? WA[k * C + li] : 0.f;            // (c): dv[p][o = li]  = sum_j du[p][j] WA[j][o]
    }
    wpa[s] = uir_split8(xa);
    wpc[s] = uir_split8(xc);
  }
  // depthwise taps of channel pair (cp, cp + 16), FLIPPED: du[p] = sum_t w[t] dvf[p - off(t)] = sum_(a,b) w[8 - (3a + b)] win[a][b]
  const int cp = tid & 15;
  const int strip = (tid >> 6) * 4 + (((lane >> 4) & 1) << 1 | (lane >> 5));     // the tile column this thread walks (0 .. 15)
  f32x2 wv[9], g[9];
  {
    const bool ok2 = cp + 16 < C;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const float u1 = a.Wdw[cp * 9 + 8 - t], u2 = a.Wdw[(ok2 ? cp + 16 : cp) * 9 + 8 - t];
      wv[t] = f32x2{u1, ok2 ? u2 : 0.f};
      g[t] = f32x2{0.f, 0.f};
    }
  }
  f32x16 gWA, gWB;
#pragma unroll
  for (int i = 0; i < 16; ++i) { gWA[i] = 0.f; gWB[i] = 0.f; }

  // interior [128][C] tiles are staged 8 lanes per pixel row of 32 floats: thread t owns float4 j = t & 7 of pixels (t >> 3) + 32 it
  const int sj = tid & 7, sp0 = tid >> 3;
  const bool sjok = 4 * sj < C;
  auto load_int = [&](float4 (&v)[T::NST], const float* base, int ld, int y0, int x0) {
#pragma unroll
    for (int it = 0; it < T::NST; ++it) {
      const int p = sp0 + 32 * it;
      const long m = img + (long)(y0 + p / TW) * a.Wd + x0 + p % TW;
      v[it] = keep4(*reinterpret_cast<const float4*>(base + m * ld + (sjok ? 4 * sj : 0)), sjok);
    }
  };
  auto store_int = [&](float* dst, const float4 (&v)[T::NST]) {
#pragma unroll
    for (int it = 0; it < T::NST; ++it) *reinterpret_cast<float4*>(dst + (sp0 + 32 * it) * T::SG + 4 * sj) = v[it];
  };
  // one per-sample Gram on the f32 MFMA: D[i][j] += sum over this wave's 32 pixels of X[p][i] Y[p][j]
  auto gram = [&](f32x16& acc, const float* X, const float* Y) {
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      const int p = 32 * wave + 2 * s + h2;
      acc = uir_mfma(X[p * T::SG + li], Y[p * T::SG + li], acc);
    }
  };

  for (int tile = w; tile < tiles_img; tile += wps) {
    const int ty = tile / tiles_x, tx = tile - ty * tiles_x;
    const int y0 = ty * T::TH, x0 = tx * TW;
    // ---- phase 0: dx1 on tile + ring as split A operands, zero outside the image; u tile in flight ----
    float4 tu[T::NST];
    load_int(tu, a.U, a.ldu, y0, x0);
    f32x16 d[T::RTW];
#pragma unroll
    for (int q = 0; q < T::RTW; ++q) {
#pragma unroll
      for (int i = 0; i < 16; ++i) d[q][i] = 0.f;
      const int r = wave + 4 * q;
      if (r < T::NRT1) {
        const int p = r * 32 + li;
        const int ry = p / T::RW, rx = p - ry * T::RW;
        const int iy = y0 + ry - 1, ix = x0 + rx - 1;
        const bool valid = p < T::NPX1 && iy >= 0 && iy < a.H && ix >= 0 && ix < a.Wd;
        const float* xr = a.dX1 + (valid ? img + (long)iy * a.Wd + ix : img) * a.ldx + 4 * h2;
        float4 xv[T::KJ];
#pragma unroll
        for (int jj = 0; jj < T::KJ; ++jj) xv[jj] = keep4(*reinterpret_cast<const float4*>(xr + 8 * jj), valid);
        // ---- (a) dvf rows of this row tile ----
#pragma unroll
        for (int s = 0; s < C / 16; ++s) d[q] = uir_mfma_split(uir_split8(xv[2 * s], xv[2 * s + 1]), wpa[s], d[q]);
      }
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
    store_int(Rs, tu);
    lds_barrier();
    // v tile in flight during the walk
    float4 tv[T::NST];
    load_int(tv, a.V, a.ldv, y0, x0);
    // ---- (b) du = transposed depthwise 3x3 of dvf; g += window x centre u ----
    {
      const float* rp = T1s + strip * T::S1 + cp;                 // ring row 0, ring column strip (= tile column strip - 1 + 1 ... + dx)
      const float* up = Rs + strip * T::SG + cp;
      f32x2 rw[3][3];
#pragma unroll
      for (int rr = 0; rr < 3; ++rr)
#pragma unroll
        for (int dx = 0; dx < 3; ++dx)
          rw[rr][dx] = f32x2{rp[(rr * T::RW + dx) * T::S1], rp[(rr * T::RW + dx) * T::S1 + 16]};
      rp += 3 * T::RW * T::S1;
      float* gdst = Gs + strip * T::SG + cp;
      auto step = [&](f32x2 (&p0)[3], f32x2 (&p1)[3], f32x2 (&p2)[3]) {
        const f32x2 uv = f32x2{up[0], up[16]};
        up += TW * T::SG;
        f32x2 t = wv[0] * p0[0];
        t = __builtin_elementwise_fma(wv[1], p0[1], t);
        t = __builtin_elementwise_fma(wv[2], p0[2], t);
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) g[kx] = __builtin_elementwise_fma(p0[kx], uv, g[kx]);
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) p0[dx] = f32x2{rp[dx * T::S1], rp[dx * T::S1 + 16]};       // ring row y + 3 (past the ring's last row: LDS rows that exist, never used)
        rp += T::RW * T::S1;
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
          t = __builtin_elementwise_fma(wv[3 + kx], p1[kx], t);
          t = __builtin_elementwise_fma(wv[6 + kx], p2[kx], t);
          g[3 + kx] = __builtin_elementwise_fma(p1[kx], uv, g[3 + kx]);
          g[6 + kx] = __builtin_elementwise_fma(p2[kx], uv, g[6 + kx]);
        }
        gdst[0] = t.x;
        gdst[16] = t.y;
        gdst += TW * T::SG;
      };
#pragma unroll 1
      for (int yb = 0; yb < T::TH; yb += 3) {
        step(rw[0], rw[1], rw[2]);
        if (yb + 1 < T::TH) step(rw[1], rw[2], rw[0]);
        if (yb + 2 < T::TH) step(rw[2], rw[0], rw[1]);
      }
    }
    lds_barrier();                      // du complete; u dead
    store_int(Rs, tv);
    float4 tf[T::NST], tx1[T::NST];     // vf and the interior of dx1 in flight during (c) and the first Gram
    load_int(tf, a.VF, a.ldf, y0, x0);
    load_int(tx1, a.dX1, a.ldx, y0, x0);
    // ---- (c) dv = du . WA: one 32-pixel row tile per wave, stored from the accumulators ----
    {
      f32x16 acc;
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[i] = 0.f;
      const float* gr = Gs + (wave * 32 + li) * T::SG + 4 * h2;
#pragma unroll
      for (int s = 0; s < C / 16; ++s)
        acc = uir_mfma_split(uir_split8(*reinterpret_cast<const float4*>(gr + 16 * s), *reinterpret_cast<const float4*>(gr + 16 * s + 8)), wpc[s], acc);
      if (cok) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int p = wave * 32 + (i & 3) + 8 * (i >> 2) + 4 * h2;
          a.dV[(img + (long)(y0 + p / TW) * a.Wd + x0 + p % TW) * a.ldd + li] = acc[i];
        }
      }
    }
    lds_barrier();                      // v staged (and every wave done with its own rows of Gs in (c): rows of one wave only)
    gram(gWA, Gs, Rs);                  // dWA[j][o] += du[p][j] v[p][o]
    lds_barrier();                      // du, v dead
    store_int(Rs, tf);
    store_int(Gs, tx1);
    lds_barrier();
    gram(gWB, Gs, Rs);                  // dWB[i][j] += dx1[p][i] vf[p][j]
    lds_barrier();                      // before the next tile's (a) / staging
  }

  // ---- the workgroup's slab rows: the waves' Gram partials and the walk's depthwise terms summed through LDS in a fixed order ----
  float* red = T1s;                     // [4 waves][2][16][64] floats = 32 KB... (T1s is 30 KB): two passes
#pragma unroll
  for (int which = 0; which < 2; ++which) {
    const f32x16& acc = which ? gWA : gWB;          // slab row = [dWB (nab) | dWA (nab)]: the layout of the three-launch path's shared slab
#pragma unroll
    for (int i = 0; i < 16; ++i) red[(wave * 16 + i) * 64 + lane] = acc[i];
    lds_barrier();
    float* row = a.slab_ab + (long)w * a.slab_ab_stride + (long)which * a.nab + (long)b * C * C;
    for (int e = tid; e < 1024; e += 256) {
      const float v = (red[e] + red[1024 + e]) + (red[2048 + e] + red[3072 + e]);
      const int i = e >> 6, ln = e & 63;
      const int m = uir_mfma_row(i, ln >> 5), n = ln & 31;
      if (m < C && n < C) row[m * C + n] = v;
    }
    lds_barrier();
  }
  {
    // depthwise terms: thread (cp, strip) holds g[3a + b] for channels cp | cp + 16; tap of window element (a, b) = 8 - (3a + b)
    float* rd = T1s;                    // [16 strips][16 cp][18]
#pragma unroll
    for (int t = 0; t < 9; ++t) { rd[(strip * 16 + cp) * 18 + t] = g[t].x; rd[(strip * 16 + cp) * 18 + 9 + t] = g[t].y; }
    lds_barrier();
    float* row = a.slab_dw + (long)blockIdx.x * (C * 9);
    for (int e = tid; e < C * 9; e += 256) {
      const int c = e / 9, t = e - 9 * c;
      const int pc = c & 15, hi = c >> 4;
      float sum = 0.f;
      for (int s = 0; s < 16; ++s) sum += rd[(s * 16 + pc) * 18 + 9 * hi + (8 - t)];
      row[e] = sum;
    }
  }
}


// ---- C = 64 (level 2): the same pass over 8 x 8 tiles, the channels of dvf / du taken 32 at a time ----
// dx1 on tile + ring stays in registers as split A operands for both chunks; per chunk: (a) dvf chunk -> LDS, (b) the walk -> du chunk, (c) dv +=
// du chunk . WA[chunk rows], (d) one 32 x 32 tile of dWA or dWB per wave over all 64 pixels of the tile (wave 0 / 1: dWA[chunk rows][32 w ..],
// wave 2 / 3: dWB[32 (w - 2) ..][chunk columns]) - every Gram tile has one owner, so the slab rows are stored from the accumulators.
struct Avb64 {
  static constexpr int C = 64, TH = 8, TW = 8, RW = TW + 2, RH = TH + 2;
  static constexpr int NPX1 = RW * RH;              // 100 tile + ring pixels: 4 row tiles of 32, one per wave
  static constexpr int NI = TH * TW;                // 64 interior pixels
  static constexpr int S1 = 40, SG = 36, SV = 68;   // LDS row strides (floats): dvf chunk on the ring / 32-channel interior rows / 64-channel interior rows
  static constexpr int OFF_G = 128 * S1;            // du chunk
  static constexpr int OFF_R = OFF_G + NI * SG;     // u chunk, then vf chunk
  static constexpr int OFF_V = OFF_R + NI * SG;     // v, all channels
  static constexpr int OFF_X = OFF_V + NI * SV;     // dx1 interior, all channels
  static constexpr int LDS_FLOATS = OFF_X + NI * SV;
};

__global__ __launch_bounds__(256, 2) void uir_attnv_bwd64_kernel(const UirAttnvBwdArgs a, const int wps, const int tiles_x, const int tiles_img) {
  using T = Avb64;
  constexpr int C = T::C, TW = T::TW;
  extern __shared__ float smem[];
  float* T1s = smem;
  float* Gs = smem + T::OFF_G;
  float* Rs = smem + T::OFF_R;
  float* Vs = smem + T::OFF_V;
  float* Xs = smem + T::OFF_X;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, h2 = lane >> 5;
  const int b = (int)blockIdx.x / wps, w = (int)blockIdx.x % wps;
  const long img = (long)b * a.H * a.Wd;
  const float* WA = a.WA + (long)b * C * C;
  const float* WB = a.WB + (long)b * C * C;
  // the walk: 16 channel pairs (cp, cp + 16 of the chunk) x 8 tile columns x 2 half heights
  const int cp = tid & 15;
  const int strip16 = (tid >> 6) * 4 + (((lane >> 4) & 1) << 1 | (lane >> 5));
  const int wcol = strip16 & 7, wrow0 = (strip16 >> 3) * 4;
  const int rt = wave & 1, nt = wave >> 1;              // (c): this wave's 32-pixel row tile and 32-column tile of dv
  // staging maps: [64][64] tiles 16 lanes per pixel row, [64][32] chunks 8 lanes per pixel row
  const int fj = tid & 15, fp0 = tid >> 4;
  const int sj = tid & 7, sp0 = tid >> 3;
  auto pix = [&](int p, int y0, int x0) { return img + (long)(y0 + p / TW) * a.Wd + x0 + p % TW; };

  for (int tile = w; tile < tiles_img; tile += wps) {
    const int ty = tile / tiles_x, tx = tile - ty * tiles_x;
    const int y0 = ty * T::TH, x0 = tx * TW;
    const bool first = tile == w;
    // ---- phase 0: v and the interior of dx1 (all channels) in flight; dx1 on tile + ring as split A operands, zero outside the image ----
    float4 tv[4], tx1[4];
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const long m = pix(fp0 + 16 * it, y0, x0);
      tv[it] = *reinterpret_cast<const float4*>(a.V + m * a.ldv + 4 * fj);
      tx1[it] = *reinterpret_cast<const float4*>(a.dX1 + m * a.ldx + 4 * fj);
    }
    UirBf3 Ap[C / 16];
    {
      const int p = wave * 32 + li;
      const int ry = p / T::RW, rx = p - ry * T::RW;
      const int iy = y0 + ry - 1, ix = x0 + rx - 1;
      const bool valid = p < T::NPX1 && iy >= 0 && iy < a.H && ix >= 0 && ix < a.Wd;
      const float* xr = a.dX1 + (valid ? img + (long)iy * a.Wd + ix : img) * a.ldx + 4 * h2;
      float4 xv[C / 8];
#pragma unroll
      for (int jj = 0; jj < C / 8; ++jj) xv[jj] = keep4(*reinterpret_cast<const float4*>(xr + 8 * jj), valid);
#pragma unroll
      for (int s = 0; s < C / 16; ++s) Ap[s] = uir_split8(xv[2 * s], xv[2 * s + 1]);
    }
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      *reinterpret_cast<float4*>(Vs + (fp0 + 16 * it) * T::SV + 4 * fj) = tv[it];
      *reinterpret_cast<float4*>(Xs + (fp0 + 16 * it) * T::SV + 4 * fj) = tx1[it];
    }
    f32x16 accv;
#pragma unroll
    for (int i = 0; i < 16; ++i) accv[i] = 0.f;

#pragma unroll 1
    for (int c = 0; c < 2; ++c) {
      const int c0 = 32 * c;
      f32x2 g[9];
#pragma unroll
      for (int t = 0; t < 9; ++t) g[t] = f32x2{0.f, 0.f};
      float4 tu[2];
#pragma unroll
      for (int it = 0; it < 2; ++it) {
        const long m = pix(sp0 + 32 * it, y0, x0);
        tu[it] = *reinterpret_cast<const float4*>(a.U + m * a.ldu + c0 + 4 * sj);
      }
      // ---- (a) dvf[p][c0 + li] = sum_i dx1[p][i] WB[i][c0 + li] on this wave's 32 ring rows ----
      {
        f32x16 d;
#pragma unroll
        for (int i = 0; i < 16; ++i) d[i] = 0.f;
#pragma unroll
        for (int s = 0; s < C / 16; ++s) {
          float xa[8];
#pragma unroll
          for (int t = 0; t < 8; ++t) xa[t] = WB[(16 * s + 8 * (t >> 2) + 4 * h2 + (t & 3)) * C + c0 + li];
          d = uir_mfma_split(Ap[s], uir_split8(xa), d);
        }
        float* dst = T1s + (wave * 32 + 4 * h2) * T::S1 + li;
#pragma unroll
        for (int i = 0; i < 16; ++i) dst[((i & 3) + 8 * (i >> 2)) * T::S1] = d[i];
      }
#pragma unroll
      for (int it = 0; it < 2; ++it) *reinterpret_cast<float4*>(Rs + (sp0 + 32 * it) * T::SG + 4 * sj) = tu[it];
      lds_barrier();
      // the vf chunk in flight during the walk
      const float4 tf0 = *reinterpret_cast<const float4*>(a.VF + pix(sp0, y0, x0) * a.ldf + c0 + 4 * sj);
      const float4 tf1 = *reinterpret_cast<const float4*>(a.VF + pix(sp0 + 32, y0, x0) * a.ldf + c0 + 4 * sj);
      // ---- (b) du chunk = transposed depthwise 3x3 of the dvf chunk; g += window x centre u ----
      {
        f32x2 wv[9];
#pragma unroll
        for (int t = 0; t < 9; ++t) wv[t] = f32x2{a.Wdw[(c0 + cp) * 9 + 8 - t], a.Wdw[(c0 + cp + 16) * 9 + 8 - t]};
        const float* rp = T1s + (wrow0 * T::RW + wcol) * T::S1 + cp;
        const float* up = Rs + (wrow0 * TW + wcol) * T::SG + cp;
        f32x2 rw[3][3];
#pragma unroll
        for (int rr = 0; rr < 3; ++rr)
#pragma unroll
          for (int dx = 0; dx < 3; ++dx)
            rw[rr][dx] = f32x2{rp[(rr * T::RW + dx) * T::S1], rp[(rr * T::RW + dx) * T::S1 + 16]};
        rp += 3 * T::RW * T::S1;
        float* gdst = Gs + (wrow0 * TW + wcol) * T::SG + cp;
        auto step = [&](f32x2 (&p0)[3], f32x2 (&p1)[3], f32x2 (&p2)[3], bool more) {
          const f32x2 uv = f32x2{up[0], up[16]};
          up += TW * T::SG;
          f32x2 t = wv[0] * p0[0];
          t = __builtin_elementwise_fma(wv[1], p0[1], t);
          t = __builtin_elementwise_fma(wv[2], p0[2], t);
#pragma unroll
          for (int kx = 0; kx < 3; ++kx) g[kx] = __builtin_elementwise_fma(p0[kx], uv, g[kx]);
          if (more) {
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) p0[dx] = f32x2{rp[dx * T::S1], rp[dx * T::S1 + 16]};
            rp += T::RW * T::S1;
          }
#pragma unroll
          for (int kx = 0; kx < 3; ++kx) {
            t = __builtin_elementwise_fma(wv[3 + kx], p1[kx], t);
            t = __builtin_elementwise_fma(wv[6 + kx], p2[kx], t);
            g[3 + kx] = __builtin_elementwise_fma(p1[kx], uv, g[3 + kx]);
            g[6 + kx] = __builtin_elementwise_fma(p2[kx], uv, g[6 + kx]);
          }
          gdst[0] = t.x;
          gdst[16] = t.y;
          gdst += TW * T::SG;
        };
        step(rw[0], rw[1], rw[2], true);
        step(rw[1], rw[2], rw[0], true);
        step(rw[2], rw[0], rw[1], true);
        step(rw[0], rw[1], rw[2], false);
      }
      lds_barrier();                      // du chunk complete; u chunk dead
      *reinterpret_cast<float4*>(Rs + sp0 * T::SG + 4 * sj) = tf0;
      *reinterpret_cast<float4*>(Rs + (sp0 + 32) * T::SG + 4 * sj) = tf1;
      {
        float* rd = T1s;                  // the walk's depthwise terms [16 strips][16 cp][18] (the dvf chunk is dead), summed after the next barrier
#pragma unroll
        for (int t = 0; t < 9; ++t) { rd[(strip16 * 16 + cp) * 18 + t] = g[t].x; rd[(strip16 * 16 + cp) * 18 + 9 + t] = g[t].y; }
      }
      // ---- (c) dv[row tile rt][32 nt + li] += du chunk . WA[c0 .. c0 + 31][32 nt + li] ----
      {
        const float* gq = Gs + (rt * 32 + li) * T::SG + 4 * h2;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          float xc[8];
#pragma unroll
          for (int t = 0; t < 8; ++t) xc[t] = WA[(c0 + 16 * s + 8 * (t >> 2) + 4 * h2 + (t & 3)) * C + 32 * nt + li];
          accv = uir_mfma_split(uir_split8(*reinterpret_cast<const float4*>(gq + 16 * s), *reinterpret_cast<const float4*>(gq + 16 * s + 8)),
                                uir_split8(xc), accv);
        }
      }
      lds_barrier();                      // vf chunk staged (v and dx1 interior since the first barrier of the tile)
      // ---- (d) this wave's Gram tile over the 64 pixels, added to the workgroup's slab row ----
      {
        f32x16 gr;
#pragma unroll
        for (int i = 0; i < 16; ++i) gr[i] = 0.f;
        const float* X = wave < 2 ? Gs + li : Xs + 32 * (wave - 2) + li;
        const float* Y = wave < 2 ? Vs + 32 * wave + li : Rs + li;
        const int sx = wave < 2 ? T::SG : T::SV, sy = wave < 2 ? T::SV : T::SG;
#pragma unroll 8
        for (int s = 0; s < 32; ++s) {
          const int p = 2 * s + h2;
          gr = uir_mfma(X[p * sx], Y[p * sy], gr);
        }
        // dWA[j = chunk row][o] (waves 0, 1) or dWB[i][j = chunk column] (waves 2, 3): one owner per element, same thread on every tile
        float* row = a.slab_ab + (long)w * a.slab_ab_stride + (long)b * C * C + (wave < 2 ? a.nab + (long)c0 * C + 32 * wave + li : (long)32 * (wave - 2) * C + c0 + li);
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          float* dst = row + uir_mfma_row(i, h2) * C;
          *dst = first ? gr[i] : *dst + gr[i];
        }
      }
      {
        const float* rd = T1s;
        float* row = a.slab_dw + (long)blockIdx.x * (C * 9) + c0 * 9;
        for (int e = tid; e < 32 * 9; e += 256) {
          const int ch = e / 9, t = e - 9 * ch;
          const int pc = ch & 15, hi = ch >> 4;
          float sum = 0.f;
#pragma unroll
          for (int s = 0; s < 16; ++s) sum += rd[(s * 16 + pc) * 18 + 9 * hi + (8 - t)];
          row[e] = first ? sum : row[e] + sum;
        }
      }
      lds_barrier();                      // before the next chunk's (a) / the next tile's staging
    }
    // dv rows of this wave's unit, from the accumulators
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int p = rt * 32 + (i & 3) + 8 * (i >> 2) + 4 * h2;
      a.dV[pix(p, y0, x0) * a.ldd + 32 * nt + li] = accv[i];
    }
  }

}

}  // namespace

extern "C" int uir_attnv_bwd_rows(int H, int Wd) {       // host helper: slab rows per image (= workgroups per image) at C <= 32, not a status
  const int tiles = (H / 8) * (Wd / 16);
  return tiles < 32 ? tiles : 32;
}
extern "C" int uir_attnv_bwd_rows_c(int H, int Wd, int C) {       // the same for any supported C (C = 64: 8 x 8 tiles, up to 64 workgroups per image)
  if (C != 64) return uir_attnv_bwd_rows(H, Wd);
  const int tiles = (H / 8) * (Wd / 8);
  return tiles < 64 ? tiles : 64;
}

extern "C" int uir_attnv_bwd(const UirAttnvBwdArgs* p, void* stream) {
  const UirAttnvBwdArgs& a = *p;
  UIR_REQUIRE(uir_fused_split() != 0);      // split-bf16 form only (engine.py keeps the three-launch path when the knob is 0)
  UIR_REQUIRE(a.dX1 && a.WA && a.WB && a.Wdw && a.U && a.VF && a.V && a.dV && a.slab_ab && a.slab_dw);
  UIR_REQUIRE(a.B > 0 && a.H > 0 && a.Wd > 0 && (a.H & 7) == 0 && (a.Wd & (a.C == 64 ? 7 : 15)) == 0 && (a.C == 16 || a.C == 32 || a.C == 64));
  UIR_REQUIRE(a.ldx >= a.C && a.ldu >= a.C && a.ldf >= a.C && a.ldv >= a.C && a.ldd >= a.C);
  UIR_REQUIRE(((a.ldx | a.ldu | a.ldf | a.ldv) & 3) == 0 && uir_aligned16(a.dX1) && uir_aligned16(a.U) && uir_aligned16(a.VF) && uir_aligned16(a.V));
  UIR_REQUIRE(a.nab == (long)a.B * a.C * a.C && a.slab_ab_stride >= 2 * a.nab);
  const int wps = uir_attnv_bwd_rows_c(a.H, a.Wd, a.C);
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (a.C == 64) {          // 8 x 8 tiles
    const int tx8 = a.Wd / 8;
    hipLaunchKernelGGL(uir_attnv_bwd64_kernel, dim3(a.B * wps), dim3(256), (size_t)Avb64::LDS_FLOATS * 4, st, a, wps, tx8, (a.H / 8) * tx8);
    return uir_launch_status();
  }
  const int tiles_x = a.Wd / 16, tiles_img = (a.H / 8) * tiles_x;
  if (a.C == 16) hipLaunchKernelGGL(uir_attnv_bwd_kernel<16>, dim3(a.B * wps), dim3(256), (size_t)AvbTile<16>::LDS_FLOATS * 4, st, a, wps, tiles_x, tiles_img);
  else hipLaunchKernelGGL(uir_attnv_bwd_kernel<32>, dim3(a.B * wps), dim3(256), (size_t)AvbTile<32>::LDS_FLOATS * 4, st, a, wps, tiles_x, tiles_img);
  return uir_launch_status();
}
