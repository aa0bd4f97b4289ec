// Input stage: Pillow-exact bilinear resize of decoded RGB bytes + ToTensor + Normalize(0.5, 0.5),
// one launch for a whole batch of images of different source sizes.
//
// Replaces transforms.Resize((1024,1024), BILINEAR) -> ToTensor -> Normalize of src/DataManipulation/DataLoader.py:13-20
// (applied per item in UIEBDataset.py:30-32), whose arithmetic is Pillow's ImagingResample for 8-bit images: integer
// work, so the bar is bit-exact.  The window/weight tables are computed on the host in double precision by
// uir_resize_coeffs (same operation order as Pillow's precompute_coeffs + normalize_coeffs_8bpc); the kernel does the
// two fixed-point passes.  One thread owns one output column of a strip of R output rows: it walks the source rows the
// strip needs once, resamples each of them horizontally to a byte (the value Pillow's intermediate image holds) and
// feeds that byte to the vertical accumulators of the output rows whose window contains the row.  The vertical
// windows and weights are uniform over the workgroup (scalar loads); the fp32 output is written once, coalesced along
// x, which is the only HBM traffic that matters (12 B per output pixel against <= 3 B per source pixel).
#include <cmath>
#include "uir_common.h"
#include "../../include/uir_kernels.h"

namespace {

constexpr int PREC = 32 - 8 - 2;      // Pillow's PRECISION_BITS for 8-bit channels
constexpr int ROWS = 16;              // output rows per workgroup strip
constexpr int KV_LDS = 128;           // vertical taps per output row kept in LDS (covers down-scaling by up to 63x)
constexpr int COLS = 64;              // output columns per workgroup: one wave, so small targets still fill the chip

__device__ __forceinline__ int clip8(int v) {
  v >>= PREC;
  return v < 0 ? 0 : (v > 255 ? 255 : v);
}

__global__ __launch_bounds__(COLS) void uir_resize_kernel(UirResizeArgs a) {
  __shared__ float lut[256];
  __shared__ int vw[ROWS * KV_LDS];
  const UirResizeImage im = a.img[blockIdx.z];
  for (int i = threadIdx.x; i < 256; i += COLS) lut[i] = a.lut[i];
  {  // the strip's vertical weights: one contiguous run of the table; read back with uniform addresses (broadcast)
    const int y0 = blockIdx.y * ROWS;
    const int n = min(ROWS, a.out_h - y0) * im.kv;
    if (im.kv <= KV_LDS)
      for (int i = threadIdx.x; i < n; i += COLS) vw[i] = a.tab[im.vk + (long)y0 * im.kv + i];
  }
  __syncthreads();

  const int* __restrict__ tab = a.tab;
  const int xx = blockIdx.x * COLS + threadIdx.x;
  const bool live = xx < a.out_w;
  const int xc = live ? xx : a.out_w - 1;
  const int y0 = blockIdx.y * ROWS;
  const int ny = min(ROWS, a.out_h - y0);

  const int xmin = tab[im.hb + 2 * xc], xn = tab[im.hb + 2 * xc + 1];
  const int* __restrict__ hk = tab + im.hk + (long)xc * im.kh;

  int vmin[ROWS], vn[ROWS];
#pragma unroll
  for (int j = 0; j < ROWS; ++j) {
    const int y = y0 + (j < ny ? j : ny - 1);
    vmin[j] = tab[im.vb + 2 * y];
    vn[j] = j < ny ? tab[im.vb + 2 * y + 1] : 0;
  }
  const int r0 = vmin[0];
  const int r1 = tab[im.vb + 2 * (y0 + ny - 1)] + tab[im.vb + 2 * (y0 + ny - 1) + 1];

  int acc[ROWS][3];
#pragma unroll
  for (int j = 0; j < ROWS; ++j) acc[j][0] = acc[j][1] = acc[j][2] = 1 << (PREC - 1);

  const unsigned char* __restrict__ src = a.src + im.src_off + (long)xmin * 3;
  const long pitch = (long)im.in_w * 3;

  // feed one horizontally resampled source row (bytes h0..h2 of this thread's column) to the vertical accumulators
  auto vertical = [&](int r, int h0, int h1, int h2) {
#pragma unroll
    for (int j = 0; j < ROWS; ++j) {
      const int t = r - vmin[j];                      // uniform over the workgroup
      if (t >= 0 && t < vn[j]) {
        const int k = im.kv <= KV_LDS ? vw[j * im.kv + t] : tab[im.vk + (long)(y0 + j) * im.kv + t];
        acc[j][0] += h0 * k;
        acc[j][1] += h1 * k;
        acc[j][2] += h2 * k;
      }
    }
  };
  // A window is a contiguous run of source bytes: four taps = 12 bytes = three unaligned dword loads per row instead
  // of twelve byte gathers (the loop is bound by load instructions and their latency), and NR source rows are
  // resampled together so 3*NR loads are in flight.  Taps past the window carry weight 0; the bytes read for them lie
  // in the next pixels / next row / the UIR_RESIZE_SRC_SLACK bytes the caller keeps readable behind the last image.
  constexpr int NR = 4;
  auto taps4 = [&](const unsigned char* __restrict__ p0, int live_rows, const int (&k)[4], int (&sum)[NR][3]) {
    unsigned w[NR][3];
#pragma unroll
    for (int i = 0; i < NR; ++i) __builtin_memcpy(w[i], p0 + (i < live_rows ? i : 0) * pitch, 12);
#pragma unroll
    for (int i = 0; i < NR; ++i)
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          const int q = 3 * u + c;
          sum[i][c] += (int)((w[i][q >> 2] >> (8 * (q & 3))) & 255u) * k[u];
        }
  };
  auto rows_out = [&](int r, int live_rows, const int (&sum)[NR][3]) {
#pragma unroll
    for (int i = 0; i < NR; ++i)
      if (i < live_rows) vertical(r + i, clip8(sum[i][0]), clip8(sum[i][1]), clip8(sum[i][2]));
  };

  if (xn <= 4 && xn > 0) {          // up-scaling and mild down-scaling: the whole window's weights in registers, once
    int k[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) k[u] = u < xn ? hk[u] : 0;
    for (int r = r0; r < r1; r += NR) {
      const int live_rows = min(NR, r1 - r);
      int sum[NR][3];
#pragma unroll
      for (int i = 0; i < NR; ++i) sum[i][0] = sum[i][1] = sum[i][2] = 1 << (PREC - 1);
      taps4(src + r * pitch, live_rows, k, sum);
      rows_out(r, live_rows, sum);
    }
  } else if (xn > 0) {
    for (int r = r0; r < r1; r += NR) {
      const int live_rows = min(NR, r1 - r);
      int sum[NR][3];
#pragma unroll
      for (int i = 0; i < NR; ++i) sum[i][0] = sum[i][1] = sum[i][2] = 1 << (PREC - 1);
      for (int t0 = 0; t0 < xn; t0 += 4) {
        int k[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) k[u] = t0 + u < xn ? hk[t0 + u] : 0;
        taps4(src + r * pitch + 3 * t0, live_rows, k, sum);
      }
      rows_out(r, live_rows, sum);
    }
  }

  if (!live) return;
  const long plane = (long)a.out_h * a.out_w;
#pragma unroll
  for (int j = 0; j < ROWS; ++j) {
    if (j < ny) {
      const int v0 = clip8(acc[j][0]), v1 = clip8(acc[j][1]), v2 = clip8(acc[j][2]);
      const long p = (long)(y0 + j) * a.out_w + xx;
      if (a.out_f32) {
        float* o = a.out_f32 + (long)blockIdx.z * 3 * plane + p;
        __builtin_nontemporal_store(lut[v0], o);             // streamed: the batch is far larger than L2
        __builtin_nontemporal_store(lut[v1], o + plane);
        __builtin_nontemporal_store(lut[v2], o + 2 * plane);
      }
      if (a.out_u8) {
        unsigned char* o = a.out_u8 + ((long)blockIdx.z * plane + p) * 3;
        o[0] = (unsigned char)v0;
        o[1] = (unsigned char)v1;
        o[2] = (unsigned char)v2;
      }
    }
  }
}

// Every window of one axis table must lie inside the source extent, and the table inside `tab`.
bool axis_ok(const int* tab, long tab_ints, int b_off, int k_off, int ksize, int out_size, int in_size) {
  if (b_off < 0 || k_off < 0 || ksize < 1) return false;
  if ((long)b_off + 2L * out_size > tab_ints || (long)k_off + (long)out_size * ksize > tab_ints) return false;
  int prev_lo = 0, prev_hi = 0;
  for (int i = 0; i < out_size; ++i) {
    const int lo = tab[b_off + 2 * i], n = tab[b_off + 2 * i + 1];
    if (lo < 0 || n < 0 || n > ksize || (long)lo + n > in_size) return false;
    if (lo < prev_lo || lo + n < prev_hi) return false;          // windows move forward with the output index
    prev_lo = lo;
    prev_hi = lo + n;
  }
  return true;
}

}  // namespace

// Host-side: the per-axis tables of Pillow's precompute_coeffs (bilinear filter, support 1, box = whole image) and
// normalize_coeffs_8bpc.  Plain double arithmetic, no contraction, in Pillow's operation order.
#pragma clang fp contract(off)
extern "C" int uir_resize_ksize(int in_size, int out_size) {
  if (in_size < 1 || out_size < 1) return -1;
  double filterscale = (double)in_size / out_size;
  if (filterscale < 1.0) filterscale = 1.0;
  const double support = 1.0 * filterscale;
  return (int)std::ceil(support) * 2 + 1;
}

extern "C" int uir_resize_coeffs(int in_size, int out_size, int* bounds, int* kk, long kk_cap) {
  const int ksize = uir_resize_ksize(in_size, out_size);
  if (ksize < 0 || !bounds || !kk || (long)out_size * ksize > kk_cap) return -1;
  const double scale = (double)in_size / out_size;
  double filterscale = scale;
  if (filterscale < 1.0) filterscale = 1.0;
  const double support = 1.0 * filterscale;
  const double ss = 1.0 / filterscale;
  double* w = new double[ksize];
  for (int xx = 0; xx < out_size; ++xx) {
    const double center = 0.0 + (xx + 0.5) * scale;
    double ww = 0.0;
    int xmin = (int)(center - support + 0.5);
    if (xmin < 0) xmin = 0;
    int xmax = (int)(center + support + 0.5);
    if (xmax > in_size) xmax = in_size;
    xmax -= xmin;
    for (int x = 0; x < xmax; ++x) {
      double v = (x + xmin - center + 0.5) * ss;
      if (v < 0.0) v = -v;
      v = v < 1.0 ? 1.0 - v : 0.0;
      w[x] = v;
      ww += v;
    }
    for (int x = 0; x < xmax; ++x)
      if (ww != 0.0) w[x] /= ww;
    int* k = kk + (long)xx * ksize;
    for (int x = 0; x < ksize; ++x) {
      const double v = x < xmax ? w[x] : 0.0;
      k[x] = v < 0 ? (int)(-0.5 + v * (1 << PREC)) : (int)(0.5 + v * (1 << PREC));
    }
    bounds[2 * xx] = xmin;
    bounds[2 * xx + 1] = xmax;
  }
  delete[] w;
  return ksize;
}

extern "C" int uir_resize_norm_u8(const UirResizeArgs* a, void* stream) {
  UIR_REQUIRE(a && a->src && a->img && a->img_host && a->tab && a->tab_host && a->lut);
  UIR_REQUIRE(a->out_f32 || a->out_u8);
  UIR_REQUIRE(a->n_img >= 1 && a->n_img <= 65535 && a->out_h >= 1 && a->out_w >= 1);
  UIR_REQUIRE((a->out_h + ROWS - 1) / ROWS <= 65535 && a->src_bytes > 0 && a->tab_ints > 0);
  UIR_REQUIRE(a->src_capacity >= a->src_bytes + UIR_RESIZE_SRC_SLACK);   // the 12-byte window loads may run past the last pixel
  for (int i = 0; i < a->n_img; ++i) {
    const UirResizeImage& im = a->img_host[i];
    UIR_REQUIRE(im.in_h >= 1 && im.in_w >= 1 && im.src_off >= 0);
    UIR_REQUIRE(im.src_off + (long)im.in_h * im.in_w * 3 <= a->src_bytes);
    if (i > 0) {   // images of one source size share their tables: checked once
      const UirResizeImage& p = a->img_host[i - 1];
      if (p.in_h == im.in_h && p.in_w == im.in_w && p.hb == im.hb && p.hk == im.hk && p.vb == im.vb && p.vk == im.vk &&
          p.kh == im.kh && p.kv == im.kv)
        continue;
    }
    UIR_REQUIRE(axis_ok(a->tab_host, a->tab_ints, im.hb, im.hk, im.kh, a->out_w, im.in_w));
    UIR_REQUIRE(axis_ok(a->tab_host, a->tab_ints, im.vb, im.vk, im.kv, a->out_h, im.in_h));
  }
  dim3 grid((a->out_w + COLS - 1) / COLS, (a->out_h + ROWS - 1) / ROWS, a->n_img);
  hipLaunchKernelGGL(uir_resize_kernel, grid, dim3(COLS), 0, (hipStream_t)stream, *a);
  return uir_launch_status();
}
