// uir_dwconv: depthwise 3x3 (pad 1, no bias) on HWC rows, HBM-bound.
// A thread owns 4 consecutive channels (one float4) of a run of R=4 pixels along W and slides
// a 3 x (R+2) window over them, so each input float4 is loaded 3*(R+2)/R = 4.5 times (from
// L1/L2) instead of 9; lanes of a wave cover consecutive channels of one pixel, i.e. every
// wave-level load/store is a contiguous C*4-byte HWC row segment.  Weights are staged once per
// workgroup into LDS, transposed to [tap][channel] (and tap-flipped for the input-gradient).
#include "uir_common.h"
#include "../../include/uir_kernels.h"

namespace {

constexpr int R = 4;

__device__ __forceinline__ float4 f4_fma(float4 a, float4 b, float4 c) {
  return make_float4(fmaf(a.x, b.x, c.x), fmaf(a.y, b.y, c.y), fmaf(a.z, b.z, c.z), fmaf(a.w, b.w, c.w));
}

__global__ __launch_bounds__(256) void uir_dwconv_kernel(const UirDwArgs a) {
  extern __shared__ __attribute__((aligned(16))) float wl[];  // [9][C]
  for (int i = threadIdx.x; i < 9 * a.C; i += blockDim.x) {
    const int t = i / a.C, c = i - t * a.C;
    const int cr = uir_colmap(c, a.grp_real, a.grp_pad);
    const int ts = a.flip ? 8 - t : t;
    wl[i] = cr >= 0 ? a.Wt[(long)cr * 9 + ts] : 0.f;
  }
  __syncthreads();
  const int c4n = a.C >> 2;
  const int runs = (a.Wd + R - 1) / R;
  const long total = (long)a.B * a.H * runs * c4n;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int c4 = (int)(idx % c4n);
    long t = idx / c4n;
    const int run = (int)(t % runs);
    t /= runs;
    const int h = (int)(t % a.H);
    const int b = (int)(t / a.H);
    const int w0 = run * R;
    float4 acc[R];
#pragma unroll
    for (int j = 0; j < R; ++j) acc[j] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int dy = -1; dy <= 1; ++dy) {
      // branch-free: clamp the neighbour coordinates, zero the out-of-image taps by select
      const int hh = h + dy;
      const bool hok = hh >= 0 && hh < a.H;
      const int hc = min(max(hh, 0), a.H - 1);
      const float* rowp = a.X + ((long)(b * a.H + hc) * a.Wd) * a.ldx + c4 * 4;
      float4 v[R + 2];
#pragma unroll
      for (int j = 0; j < R + 2; ++j) {
        const int ww = w0 - 1 + j;
        const int wc = min(max(ww, 0), a.Wd - 1);
        v[j] = *reinterpret_cast<const float4*>(rowp + (long)wc * a.ldx);
      }
#pragma unroll
      for (int j = 0; j < R + 2; ++j) {
        const int ww = w0 - 1 + j;
        if (!(hok && ww >= 0 && ww < a.Wd)) v[j] = make_float4(0.f, 0.f, 0.f, 0.f);
      }
#pragma unroll
      for (int dx = 0; dx < 3; ++dx) {
        const float4 wv = *reinterpret_cast<const float4*>(&wl[((dy + 1) * 3 + dx) * a.C + c4 * 4]);
#pragma unroll
        for (int j = 0; j < R; ++j) acc[j] = f4_fma(v[j + dx], wv, acc[j]);
      }
    }
    float* yp = a.Y + ((long)(b * a.H + h) * a.Wd) * a.ldy + c4 * 4;
#pragma unroll
    for (int j = 0; j < R; ++j)
      if (w0 + j < a.Wd) *reinterpret_cast<float4*>(yp + (long)(w0 + j) * a.ldy) = acc[j];
  }
}

// weight-gradient: thread (c4 = threadIdx.x, py = threadIdx.y) accumulates 9 float4 over its runs
__global__ __launch_bounds__(256) void uir_dwconv_wgrad_kernel(const UirDwWgradArgs a) {
  extern __shared__ __attribute__((aligned(16))) float red[];  // [ny][c4n][36]
  const int c4n = a.C >> 2;
  const int c4 = threadIdx.x, py = threadIdx.y, ny = blockDim.y;
  const int runs = (a.Wd + R - 1) / R;
  const long items = (long)a.B * a.H * runs;
  float4 acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) acc[t] = make_float4(0.f, 0.f, 0.f, 0.f);
  if (c4 < c4n) {
    for (long it = (long)blockIdx.x * ny + py; it < items; it += (long)gridDim.x * ny) {
      const int run = (int)(it % runs);
      long t = it / runs;
      const int h = (int)(t % a.H);
      const int b = (int)(t / a.H);
      const int w0 = run * R;
      float4 g[R];
      const float* gp = a.dY + ((long)(b * a.H + h) * a.Wd) * a.ldy + c4 * 4;
#pragma unroll
      for (int j = 0; j < R; ++j) g[j] = *reinterpret_cast<const float4*>(gp + (long)min(w0 + j, a.Wd - 1) * a.ldy);
#pragma unroll
      for (int j = 0; j < R; ++j)
        if (w0 + j >= a.Wd) g[j] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int dy = -1; dy <= 1; ++dy) {
        const int hh = h + dy;
        const bool hok = hh >= 0 && hh < a.H;
        const int hc = min(max(hh, 0), a.H - 1);
        const float* rowp = a.X + ((long)(b * a.H + hc) * a.Wd) * a.ldx + c4 * 4;
        float4 v[R + 2];
#pragma unroll
        for (int j = 0; j < R + 2; ++j) {
          const int wc = min(max(w0 - 1 + j, 0), a.Wd - 1);
          v[j] = *reinterpret_cast<const float4*>(rowp + (long)wc * a.ldx);
        }
#pragma unroll
        for (int j = 0; j < R + 2; ++j) {
          const int ww = w0 - 1 + j;
          if (!(hok && ww >= 0 && ww < a.Wd)) v[j] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int dx = 0; dx < 3; ++dx)
#pragma unroll
          for (int j = 0; j < R; ++j) acc[(dy + 1) * 3 + dx] = f4_fma(g[j], v[j + dx], acc[(dy + 1) * 3 + dx]);
      }
    }
    float* rp = red + ((long)py * c4n + c4) * 36;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      rp[0 * 9 + t] = acc[t].x; rp[1 * 9 + t] = acc[t].y; rp[2 * 9 + t] = acc[t].z; rp[3 * 9 + t] = acc[t].w;
    }
  }
  __syncthreads();
  // slab[block][creal][9]: sum over py in fixed order
  const int tid = threadIdx.y * blockDim.x + threadIdx.x, nthr = blockDim.x * blockDim.y;
  for (int e = tid; e < a.C * 9; e += nthr) {
    const int c = e / 9, t = e - c * 9;
    const int cr = uir_colmap(c, a.grp_real, a.grp_pad);
    if (cr < 0) continue;
    float s = 0.f;
    for (int y = 0; y < ny; ++y) s += red[((long)y * c4n + (c >> 2)) * 36 + (c & 3) * 9 + t];
    a.slab[(long)blockIdx.x * a.creal * 9 + (long)cr * 9 + t] = s;
  }
}

}  // namespace

extern "C" int uir_dwconv(const UirDwArgs* p, void* stream) {
  const UirDwArgs& a = *p;
  UIR_REQUIRE(a.X && a.Wt && a.Y);
  UIR_REQUIRE(a.B > 0 && a.H > 0 && a.Wd > 0 && a.C > 0 && (a.C & 3) == 0);
  UIR_REQUIRE((a.ldx & 3) == 0 && (a.ldy & 3) == 0 && a.ldx >= a.C && a.ldy >= a.C);
  UIR_REQUIRE(uir_aligned16(a.X) && uir_aligned16(a.Y));
  const size_t lds = (size_t)9 * a.C * sizeof(float);
  UIR_REQUIRE(lds <= 64 * 1024);
  const long total = (long)a.B * a.H * ((a.Wd + R - 1) / R) * (a.C >> 2);
  long blocks = (total + 255) / 256;
  if (blocks > 256 * 16) blocks = 256 * 16;
  hipLaunchKernelGGL(uir_dwconv_kernel, dim3((unsigned)blocks), dim3(256), lds, static_cast<hipStream_t>(stream), a);
  return uir_launch_status();
}

extern "C" int uir_dwconv_wgrad(const UirDwWgradArgs* p, void* stream) {
  const UirDwWgradArgs& a = *p;
  UIR_REQUIRE(a.dY && a.X && a.slab);
  UIR_REQUIRE(a.B > 0 && a.H > 0 && a.Wd > 0 && a.C > 0 && (a.C & 3) == 0 && a.creal > 0 && a.nblocks > 0);
  UIR_REQUIRE((a.ldx & 3) == 0 && (a.ldy & 3) == 0 && a.ldx >= a.C && a.ldy >= a.C);
  UIR_REQUIRE(uir_aligned16(a.X) && uir_aligned16(a.dY));
  const int c4n = a.C >> 2;
  UIR_REQUIRE(c4n <= 256);
  const int ny = 256 / c4n;
  const size_t lds = (size_t)ny * c4n * 36 * sizeof(float);
  hipLaunchKernelGGL(uir_dwconv_wgrad_kernel, dim3(a.nblocks), dim3(c4n, ny), lds, static_cast<hipStream_t>(stream), a);
  return uir_launch_status();
}
