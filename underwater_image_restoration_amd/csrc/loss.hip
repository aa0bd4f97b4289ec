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
    case UIR_LOSS_FFL_CHARBONNIER: loss = sums[6] + ch; break;      // sums[6] = the FocalFrequencyLoss mean (uir_ffl_partials + reduction)
    case UIR_LOSS_LUM_FFL: loss = sums[6] + lum; break;
    case UIR_LOSS_MS_SSIM: loss = 1.f - sums[7]; break;             // sums[7] = the MS-SSIM mean (uir_msssim_combine)
    case UIR_LOSS_SSIM_FFL_CHARBONNIER: loss = 0.025f * sums[6] + 0.025f * ch + (1.f - sums[7]); break;
    case UIR_LOSS_ADAPTIVE_LUM_FFL: {                               // losses.py:137-147
      const float a = 1.f / (sums[6] + 1e-6f), b = 1.f / (lum + 1e-6f);
      w0 = 0.5f * a / (a + b);
      w1 = 0.5f * b / (a + b);
      loss = sums[6] * w0 + lum * w1;
      break;
    }
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
  const bool adaptive = kind == UIR_LOSS_ADAPTIVE_LUM_CHARBONNIER || kind == UIR_LOSS_ADAPTIVE_LUM_FFL;
  const float w0 = adaptive ? aux[2] : 0.f, w1 = adaptive ? aux[3] : 0.f;
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
        // the FocalFrequencyLoss term of these three arrives separately (uir_ffl_spectrum_grad -> DFT -> uir_ffl_add_grad, accumulating)
        case UIR_LOSS_FFL_CHARBONNIER: g = d[c] / sqrtf(d[c] * d[c] + CHARB_EPS2) / nn; break;
        case UIR_LOSS_LUM_FFL: g = 2.f * lum * yc / npix; break;
        case UIR_LOSS_ADAPTIVE_LUM_FFL: g = w1 * 2.f * lum * yc / npix; break;
        case UIR_LOSS_MS_SSIM: g = 0.f; break;                     // (the MS-SSIM term arrives through uir_msssim_level_bwd + uir_axpy_dev)
        case UIR_LOSS_SSIM_FFL_CHARBONNIER: g = 0.025f * d[c] / sqrtf(d[c] * d[c] + CHARB_EPS2) / nn; break;
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

// ---- FocalFrequencyLoss (focal_frequency_loss v0.3.0 as the reference constructs it, losses.py:45-46; restated: parity unpinned) ----
__global__ void ffl_diff_kernel(const float* __restrict__ pred, const float* __restrict__ truth, float* __restrict__ d4, long HW, long n) {
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < n; idx += (long)gridDim.x * blockDim.x) {
    const long b = idx / HW, p = idx - b * HW;
    const float* pp = pred + b * 3 * HW + p;
    const float* tp = truth + b * 3 * HW + p;
    *reinterpret_cast<float4*>(d4 + idx * 4) = make_float4(pp[0] - tp[0], pp[HW] - tp[HW], pp[2 * HW] - tp[2 * HW], 0.f);
  }
}

// grid (nblk, B): squared-magnitude maxima of one image's chunk, 4 channels
__global__ __launch_bounds__(256) void ffl_max_partials_kernel(const float* __restrict__ Dr, const float* __restrict__ Di, float* __restrict__ slab,
                                                               long HW) {
  __shared__ float4 red[256];
  const long b = blockIdx.y;
  float4 m = make_float4(0.f, 0.f, 0.f, 0.f);
  for (long p = (long)blockIdx.x * 256 + threadIdx.x; p < HW; p += (long)gridDim.x * 256) {
    const float4 r = *reinterpret_cast<const float4*>(Dr + (b * HW + p) * 4), i = *reinterpret_cast<const float4*>(Di + (b * HW + p) * 4);
    m.x = fmaxf(m.x, r.x * r.x + i.x * i.x); m.y = fmaxf(m.y, r.y * r.y + i.y * i.y);
    m.z = fmaxf(m.z, r.z * r.z + i.z * i.z); m.w = fmaxf(m.w, r.w * r.w + i.w * i.w);
  }
  red[threadIdx.x] = m;
  __syncthreads();
  for (int s_ = 128; s_ > 0; s_ >>= 1) {
    if ((int)threadIdx.x < s_) {
      const float4 o = red[threadIdx.x + s_];
      float4& t = red[threadIdx.x];
      t.x = fmaxf(t.x, o.x); t.y = fmaxf(t.y, o.y); t.z = fmaxf(t.z, o.z); t.w = fmaxf(t.w, o.w);
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) *reinterpret_cast<float4*>(slab + (b * gridDim.x + blockIdx.x) * 4) = red[0];
}
__global__ void ffl_max_final_kernel(const float* __restrict__ slab, float* __restrict__ mx2, int nblk) {
  const int b = blockIdx.x, c = threadIdx.x;                 // 4 threads
  float m = 0.f;
  for (int k = 0; k < nblk; ++k) m = fmaxf(m, slab[((long)b * nblk + k) * 4 + c]);
  mx2[b * 4 + c] = m;
}

// w |D|^2 with w = |D| / max|D| per (image, channel): |D|^3 / sqrt(mx2); fixed-order sums (bitwise run-to-run)
__global__ __launch_bounds__(256) void ffl_partials_kernel(const float* __restrict__ Dr, const float* __restrict__ Di, const float* __restrict__ mx2,
                                                           float* __restrict__ slab, long HW) {
  __shared__ float red[256];
  const long b = blockIdx.y;
  const float4 mx = *reinterpret_cast<const float4*>(mx2 + b * 4);
  const float ix = mx.x > 0.f ? 1.f / sqrtf(mx.x) : 0.f, iy = mx.y > 0.f ? 1.f / sqrtf(mx.y) : 0.f, iz = mx.z > 0.f ? 1.f / sqrtf(mx.z) : 0.f;
  float acc = 0.f;
  for (long p = (long)blockIdx.x * 256 + threadIdx.x; p < HW; p += (long)gridDim.x * 256) {
    const float4 r = *reinterpret_cast<const float4*>(Dr + (b * HW + p) * 4), i = *reinterpret_cast<const float4*>(Di + (b * HW + p) * 4);
    const float a = r.x * r.x + i.x * i.x, c = r.y * r.y + i.y * i.y, e = r.z * r.z + i.z * i.z;
    acc += (a * fminf(sqrtf(a) * ix, 1.f) + c * fminf(sqrtf(c) * iy, 1.f)) + e * fminf(sqrtf(e) * iz, 1.f);
  }
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int s_ = 128; s_ > 0; s_ >>= 1) {
    if ((int)threadIdx.x < s_) red[threadIdx.x] += red[threadIdx.x + s_];
    __syncthreads();
  }
  if (threadIdx.x == 0) slab[b * gridDim.x + blockIdx.x] = red[0];
}

__global__ void ffl_spectrum_grad_kernel(const float* __restrict__ Dr, const float* __restrict__ Di, const float* __restrict__ mx2,
                                         const float* __restrict__ gscale, const float* __restrict__ wdev, float coef, float* __restrict__ Gr,
                                         float* __restrict__ Gi, long HW, long n) {
  const float k = coef * gscale[0] * (wdev ? wdev[0] : 1.f);
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < n; idx += (long)gridDim.x * blockDim.x) {
    const long b = idx / HW;
    const float4 mx = *reinterpret_cast<const float4*>(mx2 + b * 4);
    const float4 r = *reinterpret_cast<const float4*>(Dr + idx * 4), i = *reinterpret_cast<const float4*>(Di + idx * 4);
    const float wx = mx.x > 0.f ? fminf(sqrtf((r.x * r.x + i.x * i.x) / mx.x), 1.f) * k : 0.f;
    const float wy = mx.y > 0.f ? fminf(sqrtf((r.y * r.y + i.y * i.y) / mx.y), 1.f) * k : 0.f;
    const float wz = mx.z > 0.f ? fminf(sqrtf((r.z * r.z + i.z * i.z) / mx.z), 1.f) * k : 0.f;
    *reinterpret_cast<float4*>(Gr + idx * 4) = make_float4(wx * r.x, wy * r.y, wz * r.z, 0.f);
    *reinterpret_cast<float4*>(Gi + idx * 4) = make_float4(-wx * i.x, -wy * i.y, -wz * i.z, 0.f);
  }
}

__global__ void ffl_add_grad_kernel(const float* __restrict__ dd4, float* __restrict__ dpred, long HW, long n, int accumulate) {
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < n; idx += (long)gridDim.x * blockDim.x) {
    const long b = idx / HW, p = idx - b * HW;
    const float4 v = *reinterpret_cast<const float4*>(dd4 + idx * 4);
    float* o = dpred + b * 3 * HW + p;
    if (accumulate) { o[0] += v.x; o[HW] += v.y; o[2 * HW] += v.z; }
    else { o[0] = v.x; o[HW] = v.y; o[2 * HW] = v.z; }
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
  UIR_REQUIRE(sums && out && kind >= 0 && kind <= UIR_LOSS_SSIM_FFL_CHARBONNIER && B > 0 && HW > 0);
  hipLaunchKernelGGL(loss_finalize_kernel, dim3(1), dim3(64), 0, static_cast<hipStream_t>(stream), sums, out, kind, B, (long)HW);
  return uir_launch_status();
}
extern "C" int uir_loss_bwd(const float* pred, const float* truth, const float* gscale, const float* aux, float* dpred, int kind, int B, int HW,
                            void* stream) {
  UIR_REQUIRE(pred && truth && gscale && dpred && kind >= 0 && kind <= UIR_LOSS_SSIM_FFL_CHARBONNIER && kind != UIR_LOSS_GRADIENT && B > 0 && HW > 0);
  UIR_REQUIRE((kind != UIR_LOSS_ADAPTIVE_LUM_CHARBONNIER && kind != UIR_LOSS_ADAPTIVE_LUM_FFL) || aux);
  const long n = (long)B * HW;
  long blocks = (n + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(loss_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream), pred, truth, gscale, aux, dpred, kind, B, (long)HW, n);
  return uir_launch_status();
}

static unsigned ffl_grid(long n) {
  long g = (n + 255) / 256;
  return (unsigned)(g > 4096 ? 4096 : (g < 1 ? 1 : g));
}
extern "C" int uir_ffl_diff(const float* pred, const float* truth, float* d4, int B, int HW, void* stream) {
  UIR_REQUIRE(pred && truth && d4 && B > 0 && HW > 0 && uir_aligned16(d4));
  const long n = (long)B * HW;
  hipLaunchKernelGGL(ffl_diff_kernel, dim3(ffl_grid(n)), dim3(256), 0, static_cast<hipStream_t>(stream), pred, truth, d4, (long)HW, n);
  return uir_launch_status();
}
extern "C" int uir_ffl_max(const float* Dr, const float* Di, float* mx2, float* slab, int nblk, int B, int HW, void* stream) {
  UIR_REQUIRE(Dr && Di && mx2 && slab && nblk > 0 && B > 0 && HW > 0 && uir_aligned16(Dr) && uir_aligned16(Di) && uir_aligned16(slab) && uir_aligned16(mx2));
  hipLaunchKernelGGL(ffl_max_partials_kernel, dim3(nblk, B), dim3(256), 0, static_cast<hipStream_t>(stream), Dr, Di, slab, (long)HW);
  hipLaunchKernelGGL(ffl_max_final_kernel, dim3(B), dim3(4), 0, static_cast<hipStream_t>(stream), slab, mx2, nblk);
  return uir_launch_status();
}
extern "C" int uir_ffl_partials(const float* Dr, const float* Di, const float* mx2, float* slab, int nblk, int B, int HW, void* stream) {
  UIR_REQUIRE(Dr && Di && mx2 && slab && nblk > 0 && B > 0 && HW > 0 && uir_aligned16(Dr) && uir_aligned16(Di) && uir_aligned16(mx2));
  hipLaunchKernelGGL(ffl_partials_kernel, dim3(nblk, B), dim3(256), 0, static_cast<hipStream_t>(stream), Dr, Di, mx2, slab, (long)HW);
  return uir_launch_status();
}
extern "C" int uir_ffl_spectrum_grad(const float* Dr, const float* Di, const float* mx2, const float* gscale, const float* wdev, float coef,
                                     float* Gr, float* Gi, int B, int HW, void* stream) {
  UIR_REQUIRE(Dr && Di && mx2 && gscale && Gr && Gi && B > 0 && HW > 0);
  UIR_REQUIRE(uir_aligned16(Dr) && uir_aligned16(Di) && uir_aligned16(Gr) && uir_aligned16(Gi) && uir_aligned16(mx2));
  const long n = (long)B * HW;
  hipLaunchKernelGGL(ffl_spectrum_grad_kernel, dim3(ffl_grid(n)), dim3(256), 0, static_cast<hipStream_t>(stream), Dr, Di, mx2, gscale, wdev, coef,
                     Gr, Gi, (long)HW, n);
  return uir_launch_status();
}
extern "C" int uir_ffl_add_grad(const float* dd4, float* dpred, int B, int HW, int accumulate, void* stream) {
  UIR_REQUIRE(dd4 && dpred && B > 0 && HW > 0 && uir_aligned16(dd4));
  const long n = (long)B * HW;
  hipLaunchKernelGGL(ffl_add_grad_kernel, dim3(ffl_grid(n)), dim3(256), 0, static_cast<hipStream_t>(stream), dd4, dpred, (long)HW, n, accumulate);
  return uir_launch_status();
}
