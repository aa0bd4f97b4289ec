// Small kernels around the VGG16 feature stack of VGGPerceptualLoss (src/Losses/losses.py:215-255) and TVLoss (:11-28); the 3x3
// convolutions + bias + ReLU themselves are uir_gemm calls (taps = 9) on HWC rows, their input gradients uir_gemm with flipped taps
// behind uir_leaky_bwd (slope 0).  Nothing here knows the weights: the host loads a torchvision-format state dict the USER supplies
// (the pretrained file is a download the reference performs and this build cannot).
//   uir_vgg_prep / _bwd      (x - mean) / std, F.interpolate(mode="bilinear", size=(S, S), align_corners=False): NCHW3 -> HWC4 rows
//   uir_relu_inplace         nn.ReLU(inplace=True) after a convolution whose launch path has no activation epilogue
//   uir_maxpool2_hwc / _bwd  nn.MaxPool2d(2, 2) on HWC rows; the gradient goes to the first maximum of the window (scan order)
//   uir_l1_feat_partials / _bwd   F.l1_loss between two feature tensors: partial sums / sign(x - y) / n
//   uir_tv_partials / _bwd   TVLoss: sums of squared vertical / horizontal differences and their gradient
#include "uir_common.h"
#include "../../include/uir_kernels.h"

namespace {

__constant__ const float VGG_MEAN[3] = {0.485f, 0.456f, 0.406f};
__constant__ const float VGG_ISTD[3] = {1.f / 0.229f, 1.f / 0.224f, 1.f / 0.225f};

// torch's area_pixel_compute_source_index for bilinear, align_corners = False: src = max((dst + 0.5) * scale - 0.5, 0)
__device__ __forceinline__ void src_coord(int dst, float scale, int in, int& i0, int& i1, float& lam) {
  float s = ((float)dst + 0.5f) * scale - 0.5f;
  s = s < 0.f ? 0.f : s;
  i0 = (int)s;
  i0 = i0 < in - 1 ? i0 : in - 1;
  i1 = i0 < in - 1 ? i0 + 1 : i0;
  lam = s - (float)i0;
}

__global__ void vgg_prep_kernel(const float* __restrict__ img, float* __restrict__ out, int H, int W, int S, float sy, float sx, long n) {
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < n; idx += (long)gridDim.x * blockDim.x) {
    const long b = idx / ((long)S * S);
    const int r = (int)(idx - b * (long)S * S), oy = r / S, ox = r - oy * S;
    int y0, y1, x0, x1;
    float ly, lx;
    src_coord(oy, sy, H, y0, y1, ly);
    src_coord(ox, sx, W, x0, x1, lx);
    float v[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float* p = img + (b * 3 + c) * (long)H * W;
      const float top = p[(long)y0 * W + x0] * (1.f - lx) + p[(long)y0 * W + x1] * lx;
      const float bot = p[(long)y1 * W + x0] * (1.f - lx) + p[(long)y1 * W + x1] * lx;
      v[c] = ((top * (1.f - ly) + bot * ly) - VGG_MEAN[c]) * VGG_ISTD[c];
    }
    *reinterpret_cast<float4*>(out + idx * 4) = make_float4(v[0], v[1], v[2], 0.f);
  }
}

// adjoint of the above, gathered per SOURCE pixel (fixed order: no atomics): which destination rows / columns reference it
__global__ void vgg_prep_bwd_kernel(const float* __restrict__ d4, float* __restrict__ dimg, const float* __restrict__ gscale, float coef, int H,
                                    int W, int S, float sy, float sx, long n) {
  const float k = coef * gscale[0];
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < n; idx += (long)gridDim.x * blockDim.x) {
    const long b = idx / ((long)H * W);
    const int r = (int)(idx - b * (long)H * W), y = r / W, x = r - y * W;
    // candidate destinations: (dst + 0.5) * scale - 0.5 within (y - 1, y + 1)
    int dy0 = (int)floorf(((float)y - 0.5f) / sy - 0.5f) - 1, dy1 = (int)ceilf(((float)y + 1.5f) / sy - 0.5f) + 1;
    int dx0 = (int)floorf(((float)x - 0.5f) / sx - 0.5f) - 1, dx1 = (int)ceilf(((float)x + 1.5f) / sx - 0.5f) + 1;
    dy0 = dy0 < 0 ? 0 : dy0; dx0 = dx0 < 0 ? 0 : dx0;
    dy1 = dy1 > S - 1 ? S - 1 : dy1; dx1 = dx1 > S - 1 ? S - 1 : dx1;
    float acc[3] = {0.f, 0.f, 0.f};
    for (int oy = dy0; oy <= dy1; ++oy) {
      int y0, y1;
      float ly;
      src_coord(oy, sy, H, y0, y1, ly);
      const float wy = (y0 == y ? 1.f - ly : 0.f) + (y1 == y ? ly : 0.f);
      if (wy == 0.f) continue;
      for (int ox = dx0; ox <= dx1; ++ox) {
        int x0, x1;
        float lx;
        src_coord(ox, sx, W, x0, x1, lx);
        const float wx = (x0 == x ? 1.f - lx : 0.f) + (x1 == x ? lx : 0.f);
        if (wx == 0.f) continue;
        const float4 g = *reinterpret_cast<const float4*>(d4 + ((b * S + oy) * (long)S + ox) * 4);
        const float w = wy * wx;
        acc[0] = fmaf(w, g.x, acc[0]); acc[1] = fmaf(w, g.y, acc[1]); acc[2] = fmaf(w, g.z, acc[2]);
      }
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) dimg[(b * 3 + c) * (long)H * W + r] += k * acc[c] * VGG_ISTD[c];
  }
}

__global__ void maxpool2_hwc_kernel(const float* __restrict__ X, float* __restrict__ Y, int H, int W, int C4, long n) {
  const int Ho = H / 2, Wo = W / 2;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < n; idx += (long)gridDim.x * blockDim.x) {
    const int c = (int)(idx % C4);
    long t = idx / C4;
    const int ox = (int)(t % Wo);
    t /= Wo;
    const int oy = (int)(t % Ho);
    const long b = t / Ho;
    const float* p = X + (((b * H + 2 * oy) * (long)W + 2 * ox) * C4 + c) * 4;
    const float4 a = *reinterpret_cast<const float4*>(p), bq = *reinterpret_cast<const float4*>(p + C4 * 4);
    const float4 cq = *reinterpret_cast<const float4*>(p + (long)W * C4 * 4), d = *reinterpret_cast<const float4*>(p + ((long)W + 1) * C4 * 4);
    *reinterpret_cast<float4*>(Y + idx * 4) = make_float4(fmaxf(fmaxf(a.x, bq.x), fmaxf(cq.x, d.x)), fmaxf(fmaxf(a.y, bq.y), fmaxf(cq.y, d.y)),
                                                          fmaxf(fmaxf(a.z, bq.z), fmaxf(cq.z, d.z)), fmaxf(fmaxf(a.w, bq.w), fmaxf(cq.w, d.w)));
  }
}
// dX[src] (+)= dY[window of src] if src is the FIRST maximum of its window in scan order (0,0) (0,1) (1,0) (1,1)
__global__ void maxpool2_hwc_bwd_kernel(const float* __restrict__ X, const float* __restrict__ dY, float* __restrict__ dX, int H, int W, int C,
                                        long n, int accumulate) {
  const int Ho = H / 2, Wo = W / 2;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < n; idx += (long)gridDim.x * blockDim.x) {
    const int c = (int)(idx % C);
    long t = idx / C;
    const int x = (int)(t % W);
    t /= W;
    const int y = (int)(t % H);
    const long b = t / H;
    float v = 0.f;
    const int oy = y >> 1, ox = x >> 1;
    if (oy < Ho && ox < Wo) {
      const float* p = X + ((b * H + 2 * oy) * (long)W + 2 * ox) * C + c;
      const float w[4] = {p[0], p[C], p[(long)W * C], p[((long)W + 1) * C]};
      int am = 0;
#pragma unroll
      for (int i = 1; i < 4; ++i) am = w[i] > w[am] ? i : am;
      if (am == ((y & 1) << 1 | (x & 1))) v = dY[((b * Ho + oy) * (long)Wo + ox) * C + c];
    }
    dX[idx] = accumulate ? dX[idx] + v : v;
  }
}

__global__ __launch_bounds__(256) void l1_feat_partials_kernel(const float* __restrict__ X, const float* __restrict__ Y, float* __restrict__ slab, long n4) {
  __shared__ float red[256];
  float acc = 0.f;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
    const float4 a = *reinterpret_cast<const float4*>(X + i * 4), b = *reinterpret_cast<const float4*>(Y + i * 4);
    acc += (fabsf(a.x - b.x) + fabsf(a.y - b.y)) + (fabsf(a.z - b.z) + fabsf(a.w - b.w));
  }
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int s_ = 128; s_ > 0; s_ >>= 1) {
    if ((int)threadIdx.x < s_) red[threadIdx.x] += red[threadIdx.x + s_];
    __syncthreads();
  }
  if (threadIdx.x == 0) slab[blockIdx.x] = red[0];
}
__global__ void l1_feat_bwd_kernel(const float* __restrict__ X, const float* __restrict__ Y, float* __restrict__ dX, float k, long n, int accumulate) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const float d = X[i] - Y[i];
    const float g = d > 0.f ? k : (d < 0.f ? -k : 0.f);
    dX[i] = accumulate ? dX[i] + g : g;
  }
}

// TVLoss: slab[blk] = (sum of squared vertical differences, sum of squared horizontal differences) of [planes][H][W]
__global__ __launch_bounds__(256) void tv_partials_kernel(const float* __restrict__ X, float* __restrict__ slab, int H, int W, long n) {
  __shared__ float red[2][256];
  float sh = 0.f, sw = 0.f;
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < n; idx += (long)gridDim.x * 256) {
    const int r = (int)(idx % ((long)H * W)), y = r / W, x = r - y * W;
    const float v = X[idx];
    if (y + 1 < H) { const float d = X[idx + W] - v; sh = fmaf(d, d, sh); }
    if (x + 1 < W) { const float d = X[idx + 1] - v; sw = fmaf(d, d, sw); }
  }
  red[0][threadIdx.x] = sh;
  red[1][threadIdx.x] = sw;
  __syncthreads();
  for (int s_ = 128; s_ > 0; s_ >>= 1) {
    if ((int)threadIdx.x < s_) {
      red[0][threadIdx.x] += red[0][threadIdx.x + s_];
      red[1][threadIdx.x] += red[1][threadIdx.x + s_];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) { slab[blockIdx.x * 2] = red[0][0]; slab[blockIdx.x * 2 + 1] = red[1][0]; }
}
// dX += gscale * (kh * d/dx sum_h + kw * d/dx sum_w)
__global__ void tv_bwd_kernel(const float* __restrict__ X, float* __restrict__ dX, const float* __restrict__ gscale, float kh, float kw, int H, int W,
                              long n) {
  const float g = gscale[0];
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < n; idx += (long)gridDim.x * blockDim.x) {
    const int r = (int)(idx % ((long)H * W)), y = r / W, x = r - y * W;
    const float v = X[idx];
    float gh = 0.f, gw = 0.f;
    if (y + 1 < H) gh -= 2.f * (X[idx + W] - v);
    if (y > 0) gh += 2.f * (v - X[idx - W]);
    if (x + 1 < W) gw -= 2.f * (X[idx + 1] - v);
    if (x > 0) gw += 2.f * (v - X[idx - 1]);
    dX[idx] += g * (kh * gh + kw * gw);
  }
}

__global__ void relu_inplace_kernel(float* __restrict__ x, long n4) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    float4 v = *reinterpret_cast<float4*>(x + i * 4);
    v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
    *reinterpret_cast<float4*>(x + i * 4) = v;
  }
}

unsigned grid_for(long n, int cap = 8192) {
  long g = (n + 255) / 256;
  return (unsigned)(g > cap ? cap : (g < 1 ? 1 : g));
}

}  // namespace

extern "C" int uir_vgg_prep(const float* img, float* out4, int B, int H, int W, int S, void* stream) {
  UIR_REQUIRE(img && out4 && B > 0 && H > 0 && W > 0 && S > 0 && uir_aligned16(out4));
  const long n = (long)B * S * S;
  hipLaunchKernelGGL(vgg_prep_kernel, dim3(grid_for(n)), dim3(256), 0, static_cast<hipStream_t>(stream), img, out4, H, W, S, (float)H / (float)S,
                     (float)W / (float)S, n);
  return uir_launch_status();
}
extern "C" int uir_vgg_prep_bwd(const float* d4, float* dimg, const float* gscale, float coef, int B, int H, int W, int S, void* stream) {
  UIR_REQUIRE(d4 && dimg && gscale && B > 0 && H > 0 && W > 0 && S > 0 && uir_aligned16(d4));
  const long n = (long)B * H * W;
  hipLaunchKernelGGL(vgg_prep_bwd_kernel, dim3(grid_for(n)), dim3(256), 0, static_cast<hipStream_t>(stream), d4, dimg, gscale, coef, H, W, S,
                     (float)H / (float)S, (float)W / (float)S, n);
  return uir_launch_status();
}
extern "C" int uir_maxpool2_hwc(const float* X, float* Y, int B, int H, int W, int C, void* stream) {
  UIR_REQUIRE(X && Y && B > 0 && H >= 2 && W >= 2 && (H & 1) == 0 && (W & 1) == 0 && C > 0 && (C & 3) == 0 && uir_aligned16(X) && uir_aligned16(Y));
  const long n = (long)B * (H / 2) * (W / 2) * (C / 4);
  hipLaunchKernelGGL(maxpool2_hwc_kernel, dim3(grid_for(n)), dim3(256), 0, static_cast<hipStream_t>(stream), X, Y, H, W, C / 4, n);
  return uir_launch_status();
}
extern "C" int uir_maxpool2_hwc_bwd(const float* X, const float* dY, float* dX, int B, int H, int W, int C, int accumulate, void* stream) {
  UIR_REQUIRE(X && dY && dX && B > 0 && H >= 2 && W >= 2 && (H & 1) == 0 && (W & 1) == 0 && C > 0);
  const long n = (long)B * H * W * C;
  hipLaunchKernelGGL(maxpool2_hwc_bwd_kernel, dim3(grid_for(n)), dim3(256), 0, static_cast<hipStream_t>(stream), X, dY, dX, H, W, C, n, accumulate);
  return uir_launch_status();
}
extern "C" int uir_l1_feat_partials(const float* X, const float* Y, float* slab, int nblocks, long n, void* stream) {
  UIR_REQUIRE(X && Y && slab && nblocks > 0 && n > 0 && (n & 3) == 0 && uir_aligned16(X) && uir_aligned16(Y));
  hipLaunchKernelGGL(l1_feat_partials_kernel, dim3(nblocks), dim3(256), 0, static_cast<hipStream_t>(stream), X, Y, slab, n / 4);
  return uir_launch_status();
}
extern "C" int uir_l1_feat_bwd(const float* X, const float* Y, float* dX, float k, long n, int accumulate, void* stream) {
  UIR_REQUIRE(X && Y && dX && n > 0);
  hipLaunchKernelGGL(l1_feat_bwd_kernel, dim3(grid_for(n)), dim3(256), 0, static_cast<hipStream_t>(stream), X, Y, dX, k, n, accumulate);
  return uir_launch_status();
}
extern "C" int uir_tv_partials(const float* X, float* slab, int nblocks, int planes, int H, int W, void* stream) {
  UIR_REQUIRE(X && slab && nblocks > 0 && planes > 0 && H > 0 && W > 0);
  hipLaunchKernelGGL(tv_partials_kernel, dim3(nblocks), dim3(256), 0, static_cast<hipStream_t>(stream), X, slab, H, W, (long)planes * H * W);
  return uir_launch_status();
}
extern "C" int uir_tv_bwd(const float* X, float* dX, const float* gscale, float kh, float kw, int planes, int H, int W, void* stream) {
  UIR_REQUIRE(X && dX && gscale && planes > 0 && H > 0 && W > 0);
  const long n = (long)planes * H * W;
  hipLaunchKernelGGL(tv_bwd_kernel, dim3(grid_for(n)), dim3(256), 0, static_cast<hipStream_t>(stream), X, dX, gscale, kh, kw, H, W, n);
  return uir_launch_status();
}
extern "C" int uir_relu_inplace(float* x, long n, void* stream) {
  UIR_REQUIRE(x && n > 0 && (n & 3) == 0 && uir_aligned16(x));
  hipLaunchKernelGGL(relu_inplace_kernel, dim3(grid_for(n / 4)), dim3(256), 0, static_cast<hipStream_t>(stream), x, n / 4);
  return uir_launch_status();
}
