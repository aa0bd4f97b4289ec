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
  float loss, w0 = 0.f, w1 = 0.f;
  switch (kind) {
    case UIR_LOSS_L1: loss = l1 / bc; break;
    case UIR_LOSS_L2: loss = l2 / bc; break;
    case UIR_LOSS_CHARBONNIER: loss = ch; break;
    case UIR_LOSS_LUMINANCE: loss = lum; break;
    case UIR_LOSS_LUM_CHARBONNIER: loss = lum + ch; break;          // losses.py:118-123
    case UIR_LOSS_GRADIENT: loss = sums[5]; break;                  // mean |lap(d)| (uir_gradloss_partials, scaled by its reduction)
    case UIR_LOSS_ADAPTIVE_LUM_CHARBONNIER: {                       // losses.py:130-140: weights 0.5 * softmax(-log(loss_i + 1e-6)), detached
      const float a = 1.f / (ch + 1e-6f), b = 1.f / (lum + 1e-6f);  // softmax(-log x) = (1/x) / sum(1/x)
      w0 = 0.5f * a / (a + b);
      w1 = 0.5f * b / (a + b);
      loss = ch * w0 + lum * w1;
      break;
    }
    default: loss = (0.5f * l2 + 0.25f * l1 + 0.25f * lum) / bc; break;   // UIR_LOSS_L1_COLOR
  }
  out[0] = loss;
  out[1] = 20.f * log10f(1.f / sqrtf(sums[4] / n));
  out[2] = w0;
  out[3] = w1;
}

__global__ void loss_bwd_kernel(const float* __restrict__ pred, const float* __restrict__ truth, const float* __restrict__ gscale,
                                const float* __restrict__ aux, float* __restrict__ dpred, int kind, int B, long HW, long n) {
  const float gs = gscale[0];
  const float w0 = kind == UIR_LOSS_ADAPTIVE_LUM_CHARBONNIER ? aux[2] : 0.f, w1 = kind == UIR_LOSS_ADAPTIVE_LUM_CHARBONNIER ? aux[3] : 0.f;
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
        case UIR_LOSS_ADAPTIVE_LUM_CHARBONNIER: g = w1 * 2.f * lum * yc / npix + w0 * d[c] / sqrtf(d[c] * d[c] + CHARB_EPS2) / nn; break;
        default: g = (0.5f * 2.f * d[c] / nn + 0.25f * sg / nn + 0.25f * 2.f * lum * yc / npix) / bc; break;
      }
      dpred[b * 3 * HW + c * HW + p] = g * gs;
    }
  }
}

// ---- Gradient_Loss (losses.py:162-181).  lap(d)[y][x] for 1 <= y <= H-2, 1 <= x <= W-2 (the valid convolution's outputs) ----
__device__ __forceinline__ float lap_at(const float* __restrict__ p, const float* __restrict__ t, long o, int Wd) {
  return (p[o - Wd] - t[o - Wd]) + (p[o - 1] - t[o - 1]) + (p[o + 1] - t[o + 1]) + (p[o + Wd] - t[o + Wd]) - 4.f * (p[o] - t[o]);
}

__global__ __launch_bounds__(256) void gradloss_partials_kernel(const float* __restrict__ pred, const float* __restrict__ truth,
                                                                float* __restrict__ slab, int H, int Wd, long planes) {
  __shared__ float red[4];
  const int iw = Wd - 2, ih = H - 2;
  const long n = planes * ih * iw;
  float s = 0.f;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < n; idx += (long)gridDim.x * blockDim.x) {
    const long pl = idx / ((long)ih * iw), r = idx - pl * ih * iw;
    const int y = (int)(r / iw) + 1, x = (int)(r - (long)(y - 1) * iw) + 1;
    s += fabsf(lap_at(pred, truth, pl * H * Wd + (long)y * Wd + x, Wd));
  }
  s = uir_wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) slab[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ __launch_bounds__(256) void gradloss_bwd_kernel(const float* __restrict__ pred, const float* __restrict__ truth,
                                                           const float* __restrict__ gscale, float* __restrict__ dpred, int H, int Wd,
                                                           long planes) {
  // d loss / d d[q] = 1/N * sum over the outputs o whose stencil holds q of sign(lap(d)[o]) * k[q - o]
  const float sc = gscale[0] / ((float)planes * (float)(H - 2) * (float)(Wd - 2));
  const long n = planes * H * Wd;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < n; idx += (long)gridDim.x * blockDim.x) {
    const long pl = idx / ((long)H * Wd), r = idx - pl * H * Wd;
    const int y = (int)(r / Wd), x = (int)(r - (long)y * Wd);
    const long base = pl * H * Wd;
    auto sg = [&](int oy, int ox) -> float {      // sign of lap(d) at output (oy, ox), 0 outside the valid range
      if (oy < 1 || oy > H - 2 || ox < 1 || ox > Wd - 2) return 0.f;
      const float v = lap_at(pred, truth, base + (long)oy * Wd + ox, Wd);
      return v > 0.f ? 1.f : (v < 0.f ? -1.f : 0.f);
    };
    dpred[idx] = sc * (sg(y - 1, x) + sg(y + 1, x) + sg(y, x - 1) + sg(y, x + 1) - 4.f * sg(y, x));
  }
}

}  // namespace

extern "C" int uir_gradloss_partials(const float* pred, const float* truth, float* slab, int nblocks, int B, int H, int Wd, void* stream) {
  UIR_REQUIRE(pred && truth && slab && nblocks > 0 && B > 0 && H >= 3 && Wd >= 3);
  hipLaunchKernelGGL(gradloss_partials_kernel, dim3(nblocks), dim3(256), 0, static_cast<hipStream_t>(stream), pred, truth, slab, H, Wd, (long)B * 3);
  return uir_launch_status();
}
extern "C" int uir_gradloss_bwd(const float* pred, const float* truth, const float* gscale, float* dpred, int B, int H, int Wd, void* stream) {
  UIR_REQUIRE(pred && truth && gscale && dpred && B > 0 && H >= 3 && Wd >= 3);
  const long n = (long)B * 3 * H * Wd;
  long blocks = (n + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(gradloss_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream), pred, truth, gscale, dpred, H, Wd, (long)B * 3);
  return uir_launch_status();
}

extern "C" int uir_loss_partials(const float* pred, const float* truth, float* slab, int nblocks, int B, int HW, void* stream) {
  UIR_REQUIRE(pred && truth && slab && nblocks > 0 && B > 0 && HW > 0);
  hipLaunchKernelGGL(loss_partials_kernel, dim3(nblocks), dim3(256), 0, static_cast<hipStream_t>(stream), pred, truth, slab, (long)HW, (long)B * HW);
  return uir_launch_status();
}
extern "C" int uir_loss_finalize(const float* sums, float* out, int kind, int B, int HW, void* stream) {
  UIR_REQUIRE(sums && out && kind >= 0 && kind <= UIR_LOSS_GRADIENT && B > 0 && HW > 0);
  hipLaunchKernelGGL(loss_finalize_kernel, dim3(1), dim3(64), 0, static_cast<hipStream_t>(stream), sums, out, kind, B, (long)HW);
  return uir_launch_status();
}
extern "C" int uir_loss_bwd(const float* pred, const float* truth, const float* gscale, const float* aux, float* dpred, int kind, int B, int HW,
                            void* stream) {
  UIR_REQUIRE(pred && truth && gscale && dpred && kind >= 0 && kind <= UIR_LOSS_ADAPTIVE_LUM_CHARBONNIER && B > 0 && HW > 0);
  UIR_REQUIRE(kind != UIR_LOSS_ADAPTIVE_LUM_CHARBONNIER || aux);
  const long n = (long)B * HW;
  long blocks = (n + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(loss_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream), pred, truth, gscale, aux, dpred, kind, B, (long)HW, n);
  return uir_launch_status();
}
