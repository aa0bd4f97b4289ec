// Loss reductions and their gradients on NCHW [B,3,H,W] prediction / target pairs
// (src/Losses/losses.py:55-57,76-78,190-192,209-212; luminanceLoss.py:16-19) and the PSNR
// reduction (src/ModelTrainer.py:17-21).  One pass produces every partial sum any of the
// hot-path losses needs; slabs are summed in fixed order by uir_reduce_slab.
#include "uir_common.h"
#include "../../include/uir_kernels.h"

namespace {

constexpr float YR = 0.299f, YG = 0.587f, YB = 0.114f;
constexpr float CHARB_EPS2 = 1e-3f * 1e-3f;

__device__ __forceinline__ float clamp01(float v) { return fminf(fmaxf(v, 0.f), 1.f); }

__global__ __launch_bounds__(256) void loss_partials_kernel(const float* __restrict__ pred, const float* __restrict__ truth,
                                                            float* __restrict__ slab, long HW, long n) {
  __shared__ float red[4][5];
  float s_abs = 0.f, s_sq = 0.f, s_ch = 0.f, s_lum = 0.f, s_cl = 0.f;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < n; idx += (long)gridDim.x * blockDim.x) {
    const long b = idx / HW, p = idx - b * HW;
    const float* pp = pred + b * 3 * HW + p;
    const float* tp = truth + b * 3 * HW + p;
    float lum = 0.f;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float pv = pp[c * HW], tv = tp[c * HW];
      const float d = pv - tv;
      s_abs += fabsf(d);
      s_sq += d * d;
      s_ch += sqrtf(d * d + CHARB_EPS2);
      lum += (c == 0 ? YR : (c == 1 ? YG : YB)) * d;
      const float dc = clamp01(pv) - clamp01(tv);
      s_cl += dc * dc;
    }
    s_lum += lum * lum;
  }
  s_abs = uir_wave_sum(s_abs); s_sq = uir_wave_sum(s_sq); s_ch = uir_wave_sum(s_ch);
  s_lum = uir_wave_sum(s_lum); s_cl = uir_wave_sum(s_cl);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) { red[wave][0] = s_abs; red[wave][1] = s_sq; red[wave][2] = s_ch; red[wave][3] = s_lum; red[wave][4] = s_cl; }
  __syncthreads();
  if (threadIdx.x < 8) {
    float s = 0.f;
    if (threadIdx.x < 5) for (int w = 0; w < 4; ++w) s += red[w][threadIdx.x];
    slab[(long)blockIdx.x * 8 + threadIdx.x] = s;
  }
}

__global__ void loss_finalize_kernel(const float* __restrict__ sums, float* __restrict__ out, int kind, int B, long HW) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  const float n = (float)B * 3.f * (float)HW;
  const float npix = (float)B * (float)HW;
  const float bc = (float)B * 3.f;
  const float l1 = sums[0] / n, l2 = sums[1] / n, ch = sums[2] / n, lum = sums[3] / npix;
  float loss;
  switch (kind) {
    case UIR_LOSS_L1: loss = l1 / bc; break;
    case UIR_LOSS_L2: loss = l2 / bc; break;
    case UIR_LOSS_CHARBONNIER: loss = ch; break;
    case UIR_LOSS_LUMINANCE: loss = lum; break;
    case UIR_LOSS_LUM_CHARBONNIER: loss = lum + ch; break;          // losses.py:118-123
    default: loss = (0.5f * l2 + 0.25f * l1 + 0.25f * lum) / bc; break;   // UIR_LOSS_L1_COLOR
  }
  out[0] = loss;
  out[1] = 20.f * log10f(1.f / sqrtf(sums[4] / n));
}

__global__ void loss_bwd_kernel(const float* __restrict__ pred, const float* __restrict__ truth, const float* __restrict__ gscale,
                                float* __restrict__ dpred, int kind, int B, long HW, long n) {
  const float gs = gscale[0];
  const float nn = (float)B * 3.f * (float)HW;
  const float npix = (float)B * (float)HW;
  const float bc = (float)B * 3.f;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < n; idx += (long)gridDim.x * blockDim.x) {
    const long b = idx / HW, p = idx - b * HW;
    const float* pp = pred + b * 3 * HW + p;
    const float* tp = truth + b * 3 * HW + p;
    float d[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) d[c] = pp[c * HW] - tp[c * HW];
    const float lum = YR * d[0] + YG * d[1] + YB * d[2];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float yc = (c == 0 ? YR : (c == 1 ? YG : YB));
      const float sg = d[c] > 0.f ? 1.f : (d[c] < 0.f ? -1.f : 0.f);
      float g;
      switch (kind) {
        case UIR_LOSS_L1: g = sg / (nn * bc); break;
        case UIR_LOSS_L2: g = 2.f * d[c] / (nn * bc); break;
        case UIR_LOSS_CHARBONNIER: g = d[c] / sqrtf(d[c] * d[c] + CHARB_EPS2) / nn; break;
        case UIR_LOSS_LUMINANCE: g = 2.f * lum * yc / npix; break;
        case UIR_LOSS_LUM_CHARBONNIER: g = 2.f * lum * yc / npix + d[c] / sqrtf(d[c] * d[c] + CHARB_EPS2) / nn; break;
        default: g = (0.5f * 2.f * d[c] / nn + 0.25f * sg / nn + 0.25f * 2.f * lum * yc / npix) / bc; break;
      }
      dpred[b * 3 * HW + c * HW + p] = g * gs;
    }
  }
}

}  // namespace

extern "C" int uir_loss_partials(const float* pred, const float* truth, float* slab, int nblocks, int B, int HW, void* stream) {
  UIR_REQUIRE(pred && truth && slab && nblocks > 0 && B > 0 && HW > 0);
  hipLaunchKernelGGL(loss_partials_kernel, dim3(nblocks), dim3(256), 0, static_cast<hipStream_t>(stream), pred, truth, slab, (long)HW, (long)B * HW);
  return uir_launch_status();
}
extern "C" int uir_loss_finalize(const float* sums, float* out, int kind, int B, int HW, void* stream) {
  UIR_REQUIRE(sums && out && kind >= 0 && kind <= UIR_LOSS_LUM_CHARBONNIER && B > 0 && HW > 0);
  hipLaunchKernelGGL(loss_finalize_kernel, dim3(1), dim3(64), 0, static_cast<hipStream_t>(stream), sums, out, kind, B, (long)HW);
  return uir_launch_status();
}
extern "C" int uir_loss_bwd(const float* pred, const float* truth, const float* gscale, float* dpred, int kind, int B, int HW, void* stream) {
  UIR_REQUIRE(pred && truth && gscale && dpred && kind >= 0 && kind <= UIR_LOSS_LUM_CHARBONNIER && B > 0 && HW > 0);
  const long n = (long)B * HW;
  long blocks = (n + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(loss_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream), pred, truth, gscale, dpred, kind, B, (long)HW, n);
  return uir_launch_status();
}
