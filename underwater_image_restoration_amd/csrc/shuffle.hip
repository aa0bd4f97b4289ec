// Index-permutation kernels on HWC rows: the zero-interleave of the spectral up-sampler,
// PixelShuffle / PixelUnshuffle, channel-slice copies (torch.cat joins) and the NCHW<->HWC
// conversion at the model boundary.  Pure data movement, one element per thread, channel
// index fastest so every wave access is a contiguous channel run.
#include "uir_common.h"
#include "../../include/uir_kernels.h"

namespace {

inline dim3 ew_grid(long n) {
  long b = (n + 255) / 256;
  if (b > 16384) b = 16384;
  if (b < 1) b = 1;
  return dim3((unsigned)b);
}

#define GRID_STRIDE(idx, n) \
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < (n); idx += (long)gridDim.x * blockDim.x)

__global__ void scatter_even_kernel(const float* __restrict__ src, int lds, const float* __restrict__ fill,
                                    float* __restrict__ dst, int ldd, int dcol, int H, int Wd, int C, long n) {
  GRID_STRIDE(idx, n) {  // idx over dst [B][2H][2W][C]
    const int c = (int)(idx % C);
    long p = idx / C;
    const int w2 = (int)(p % (2 * Wd));
    p /= (2 * Wd);
    const int h2 = (int)(p % (2 * H));
    const long b = p / (2 * H);
    float v;
    if (((w2 | h2) & 1) == 0) v = src[((b * H + (h2 >> 1)) * Wd + (w2 >> 1)) * lds + c];
    else v = fill ? fill[c] : 0.f;
    dst[((b * 2 * H + h2) * 2 * Wd + w2) * ldd + dcol + c] = v;
  }
}

__global__ void gather_even_kernel(const float* __restrict__ src, int lds, int scol, float* __restrict__ dst, int ldd,
                                   int H, int Wd, int C, long n) {
  GRID_STRIDE(idx, n) {  // idx over dst [B][H][W][C]
    const int c = (int)(idx % C);
    long p = idx / C;
    const int w = (int)(p % Wd);
    p /= Wd;
    const int h = (int)(p % H);
    const long b = p / H;
    dst[((b * H + h) * Wd + w) * ldd + c] = src[((b * 2 * H + 2 * h) * 2 * Wd + 2 * w) * lds + scol + c];
  }
}

__global__ void pixel_shuffle_kernel(const float* __restrict__ src, int lds, float* __restrict__ dst, int ldd, int dcol,
                                     int H, int Wd, int C, long n) {
  GRID_STRIDE(idx, n) {  // idx over dst [B][2H][2W][C]
    const int c = (int)(idx % C);
    long p = idx / C;
    const int w2 = (int)(p % (2 * Wd));
    p /= (2 * Wd);
    const int h2 = (int)(p % (2 * H));
    const long b = p / (2 * H);
    const int sub = (h2 & 1) * 2 + (w2 & 1);
    dst[((b * 2 * H + h2) * 2 * Wd + w2) * ldd + dcol + c] =
        src[((b * H + (h2 >> 1)) * Wd + (w2 >> 1)) * lds + c * 4 + sub];
  }
}

__global__ void pixel_unshuffle_kernel(const float* __restrict__ src, int lds, int scol, float* __restrict__ dst, int ldd,
                                       int H, int Wd, int C, long n) {
  GRID_STRIDE(idx, n) {  // idx over dst [B][H][W][4C]
    const int cc = (int)(idx % (4 * C));
    long p = idx / (4 * C);
    const int w = (int)(p % Wd);
    p /= Wd;
    const int h = (int)(p % H);
    const long b = p / H;
    const int c = cc >> 2, sub = cc & 3;
    dst[((b * H + h) * Wd + w) * ldd + cc] =
        src[((b * 2 * H + 2 * h + (sub >> 1)) * 2 * Wd + 2 * w + (sub & 1)) * lds + scol + c];
  }
}

__global__ void copy_cols_kernel(const float* __restrict__ src, int lds, int scol, float* __restrict__ dst, int ldd, int dcol,
                                 int C, int accumulate, long n) {
  GRID_STRIDE(idx, n) {
    const int c = (int)(idx % C);
    const long m = idx / C;
    const float v = src[m * lds + scol + c];
    float* d = dst + m * ldd + dcol + c;
    *d = accumulate ? *d + v : v;
  }
}

__global__ void nchw3_to_hwc4_kernel(const float* __restrict__ src, float* __restrict__ dst, long HW, long n) {
  GRID_STRIDE(idx, n) {  // idx over [B][HW]
    const long b = idx / HW, p = idx - b * HW;
    const float* s = src + b * 3 * HW + p;
    *reinterpret_cast<float4*>(dst + idx * 4) = make_float4(s[0], s[HW], s[2 * HW], 0.f);
  }
}

// 3x3 patches of the 3-channel NCHW input as HWC rows of 28 floats: column cin*9 + tap (= the Conv2d weight's own
// [cin][ky][kx] order), zero outside the image, column 27 zero.  Turns embed_conv_rgb (SpectralTransformer.py:219,254)
// and its weight gradient into a plain K = 27 GEMM / Gram instead of nine 4-channel taps on 32x32 MFMA tiles.
__global__ void im2col3x3_c3_kernel(const float* __restrict__ src, float* __restrict__ dst, int H, int Wd, long HW, long n) {
  GRID_STRIDE(idx, n) {  // idx over [B][HW]
    const long b = idx / HW, p = idx - b * HW;
    const int y = (int)(p / Wd), x = (int)(p - (long)y * Wd);
    float v[28];
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        const int yy = y + t / 3 - 1, xx = x + t % 3 - 1;
        const bool in = yy >= 0 && yy < H && xx >= 0 && xx < Wd;
        const float u = src[(b * 3 + c) * HW + (in ? (long)yy * Wd + xx : p)];
        v[c * 9 + t] = in ? u : 0.f;
      }
    v[27] = 0.f;
    float4* d = reinterpret_cast<float4*>(dst + idx * 28);
#pragma unroll
    for (int q = 0; q < 7; ++q) d[q] = make_float4(v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]);
  }
}

__global__ void hwc4_to_nchw3_kernel(const float* __restrict__ src, float* __restrict__ dst, long HW, long n) {
  GRID_STRIDE(idx, n) {
    const long b = idx / HW, p = idx - b * HW;
    const float4 v = *reinterpret_cast<const float4*>(src + idx * 4);
    float* d = dst + b * 3 * HW + p;
    d[0] = v.x; d[HW] = v.y; d[2 * HW] = v.z;
  }
}

// dst[i] = sum_{g < group} src[idx[i * group + g]] (entries with idx < 0 contribute 0): weight re-layouts of the AST samplers
__global__ void gather_sum_kernel(const float* __restrict__ src, const int* __restrict__ idx, float* __restrict__ dst, long n, int group) {
  GRID_STRIDE(i, n) {
    float acc = 0.f;
    for (int g = 0; g < group; ++g) {
      const int j = idx[i * group + g];
      acc += j >= 0 ? src[j] : 0.f;
    }
    dst[i] = acc;
  }
}

}  // namespace

extern "C" int uir_gather_sum(const float* src, const int* idx, float* dst, long n, int group, void* stream) {
  UIR_REQUIRE(src && idx && dst && n > 0 && group > 0);
  hipLaunchKernelGGL(gather_sum_kernel, ew_grid(n), dim3(256), 0, static_cast<hipStream_t>(stream), src, idx, dst, n, group);
  return uir_launch_status();
}
extern "C" int uir_scatter_even(const float* src, int lds, const float* fill, float* dst, int ldd, int dcol,
                                int B, int H, int Wd, int C, void* stream) {
  UIR_REQUIRE(src && dst && B > 0 && H > 0 && Wd > 0 && C > 0 && lds >= C && ldd >= dcol + C);
  const long n = (long)B * 4 * H * Wd * C;
  hipLaunchKernelGGL(scatter_even_kernel, ew_grid(n), dim3(256), 0, static_cast<hipStream_t>(stream), src, lds, fill, dst, ldd, dcol, H, Wd, C, n);
  return uir_launch_status();
}
extern "C" int uir_gather_even(const float* src, int lds, int scol, float* dst, int ldd, int B, int H, int Wd, int C, void* stream) {
  UIR_REQUIRE(src && dst && B > 0 && H > 0 && Wd > 0 && C > 0 && lds >= scol + C && ldd >= C);
  const long n = (long)B * H * Wd * C;
  hipLaunchKernelGGL(gather_even_kernel, ew_grid(n), dim3(256), 0, static_cast<hipStream_t>(stream), src, lds, scol, dst, ldd, H, Wd, C, n);
  return uir_launch_status();
}
extern "C" int uir_pixel_shuffle(const float* src, int lds, float* dst, int ldd, int dcol, int B, int H, int Wd, int C, void* stream) {
  UIR_REQUIRE(src && dst && B > 0 && H > 0 && Wd > 0 && C > 0 && lds >= 4 * C && ldd >= dcol + C);
  const long n = (long)B * 4 * H * Wd * C;
  hipLaunchKernelGGL(pixel_shuffle_kernel, ew_grid(n), dim3(256), 0, static_cast<hipStream_t>(stream), src, lds, dst, ldd, dcol, H, Wd, C, n);
  return uir_launch_status();
}
extern "C" int uir_pixel_unshuffle(const float* src, int lds, int scol, float* dst, int ldd, int B, int H, int Wd, int C, void* stream) {
  UIR_REQUIRE(src && dst && B > 0 && H > 0 && Wd > 0 && C > 0 && lds >= scol + C && ldd >= 4 * C);
  const long n = (long)B * H * Wd * 4 * C;
  hipLaunchKernelGGL(pixel_unshuffle_kernel, ew_grid(n), dim3(256), 0, static_cast<hipStream_t>(stream), src, lds, scol, dst, ldd, H, Wd, C, n);
  return uir_launch_status();
}
extern "C" int uir_copy_cols(const float* src, int lds, int scol, float* dst, int ldd, int dcol, long M, int C, int accumulate, void* stream) {
  UIR_REQUIRE(src && dst && M > 0 && C > 0 && lds >= scol + C && ldd >= dcol + C);
  const long n = M * C;
  hipLaunchKernelGGL(copy_cols_kernel, ew_grid(n), dim3(256), 0, static_cast<hipStream_t>(stream), src, lds, scol, dst, ldd, dcol, C, accumulate, n);
  return uir_launch_status();
}
extern "C" int uir_nchw3_to_hwc4(const float* src, float* dst, int B, int H, int Wd, void* stream) {
  UIR_REQUIRE(src && dst && B > 0 && H > 0 && Wd > 0 && uir_aligned16(dst));
  const long HW = (long)H * Wd, n = (long)B * HW;
  hipLaunchKernelGGL(nchw3_to_hwc4_kernel, ew_grid(n), dim3(256), 0, static_cast<hipStream_t>(stream), src, dst, HW, n);
  return uir_launch_status();
}
extern "C" int uir_im2col3x3_c3(const float* src, float* dst, int B, int H, int Wd, void* stream) {
  UIR_REQUIRE(src && dst && B > 0 && H > 0 && Wd > 0 && uir_aligned16(dst));
  const long HW = (long)H * Wd, n = (long)B * HW;
  hipLaunchKernelGGL(im2col3x3_c3_kernel, ew_grid(n), dim3(256), 0, static_cast<hipStream_t>(stream), src, dst, H, Wd, HW, n);
  return uir_launch_status();
}
extern "C" int uir_hwc4_to_nchw3(const float* src, float* dst, int B, int H, int Wd, void* stream) {
  UIR_REQUIRE(src && dst && B > 0 && H > 0 && Wd > 0 && uir_aligned16(src));
  const long HW = (long)H * Wd, n = (long)B * HW;
  hipLaunchKernelGGL(hwc4_to_nchw3_kernel, ew_grid(n), dim3(256), 0, static_cast<hipStream_t>(stream), src, dst, HW, n);
  return uir_launch_status();
}
