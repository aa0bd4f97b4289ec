// uir_msssim_*: the pieces of MS_SSIM(win_size=11, win_sigma=1.5, data_range=1, size_average=True, channel=3) as a TRAINING loss
// (src/Losses/losses.py:43-44: "ms_ssim" = 1 - MS_SSIM (:86-87), "ssimFFLCharbonnier" (:152-156)), forward and backward.
// pytorch_msssim is an un-vendored, unpinned pip package that is not in this image: its published algorithm is restated
// (PARITY UNPINNED; checked against oracle/msssim_ref.py):
//   per level (5 of them): mu = G * x (separable VALID 11-tap Gaussian = the outer-product window), sigma terms from G * x^2, G * y^2,
//   G * xy;  cs_map = (2 s12 + C2) / (s1 + s2 + C2),  ssim_map = (2 mu1 mu2 + C1) / (mu1^2 + mu2^2 + C1) * cs_map;  their spatial means
//   per (image, channel);  between levels avg_pool2d(kernel 2, padding = size % 2, zeros counted);
//   ms_ssim = mean over (image, channel) of  prod_{i<4} relu(cs_i)^w_i * relu(ssim_4)^w_4.
// The loss sees 3-channel images (3 M pixels at 16 x 256 x 256 against the network's 1 G activations): the kernels are plain - one
// thread per output, the 121-tap window read through the caches - and sum in a fixed order (bitwise run-to-run).
#include "uir_common.h"
#include "../../include/uir_kernels.h"

namespace {

constexpr int WIN = 11;

struct Moments { float mu1, mu2, xx, yy, xy; };

// the five windowed moments at valid position (oy, ox) of one plane
__device__ __forceinline__ Moments moments_at(const float* __restrict__ X, const float* __restrict__ Y, int W, int oy, int ox,
                                              const float (&g)[WIN]) {
  Moments m{0.f, 0.f, 0.f, 0.f, 0.f};
  for (int dy = 0; dy < WIN; ++dy) {
    float r1 = 0.f, r2 = 0.f, rxx = 0.f, ryy = 0.f, rxy = 0.f;
    const float* xp = X + (long)(oy + dy) * W + ox;
    const float* yp = Y + (long)(oy + dy) * W + ox;
#pragma unroll
    for (int dx = 0; dx < WIN; ++dx) {
      const float x = xp[dx], y = yp[dx], w = g[dx];
      r1 = fmaf(w, x, r1); r2 = fmaf(w, y, r2);
      rxx = fmaf(w, x * x, rxx); ryy = fmaf(w, y * y, ryy); rxy = fmaf(w, x * y, rxy);
    }
    const float wy = g[dy];
    m.mu1 = fmaf(wy, r1, m.mu1); m.mu2 = fmaf(wy, r2, m.mu2);
    m.xx = fmaf(wy, rxx, m.xx); m.yy = fmaf(wy, ryy, m.yy); m.xy = fmaf(wy, rxy, m.xy);
  }
  return m;
}

// grid (nblk, planes): partial sums of ssim_map and cs_map over the plane's valid positions
__global__ __launch_bounds__(256) void msssim_partials_kernel(const UirMsSsimArgs a) {
  __shared__ float red[2][256];
  const int plane = blockIdx.y;
  const int Ho = a.H - (WIN - 1), Wo = a.W - (WIN - 1);
  const long npos = (long)Ho * Wo;
  const float* X = a.X + (long)plane * a.H * a.W;
  const float* Y = a.Y + (long)plane * a.H * a.W;
  float g[WIN];
#pragma unroll
  for (int i = 0; i < WIN; ++i) g[i] = a.win[i];
  float ss = 0.f, cs = 0.f;
  for (long p = (long)blockIdx.x * 256 + threadIdx.x; p < npos; p += (long)gridDim.x * 256) {
    const int oy = (int)(p / Wo), ox = (int)(p - (long)oy * Wo);
    const Moments m = moments_at(X, Y, a.W, oy, ox, g);
    const float s1 = m.xx - m.mu1 * m.mu1, s2 = m.yy - m.mu2 * m.mu2, s12 = m.xy - m.mu1 * m.mu2;
    const float c = (2.f * s12 + a.C2) / (s1 + s2 + a.C2);
    const float l = (2.f * m.mu1 * m.mu2 + a.C1) / (m.mu1 * m.mu1 + m.mu2 * m.mu2 + a.C1);
    ss += l * c;
    cs += c;
  }
  red[0][threadIdx.x] = ss;
  red[1][threadIdx.x] = cs;
  __syncthreads();
  for (int s_ = 128; s_ > 0; s_ >>= 1) {
    if ((int)threadIdx.x < s_) {
      red[0][threadIdx.x] += red[0][threadIdx.x + s_];
      red[1][threadIdx.x] += red[1][threadIdx.x + s_];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    a.slab[((long)plane * gridDim.x + blockIdx.x) * 2] = red[0][0];
    a.slab[((long)plane * gridDim.x + blockIdx.x) * 2 + 1] = red[1][0];
  }
}
// means[plane] = (mean ssim_map, mean cs_map)
__global__ void msssim_means_kernel(const float* __restrict__ slab, float* __restrict__ means, int nblk, float inv_npos) {
  const int plane = blockIdx.x, which = threadIdx.x;            // 2 threads
  float s = 0.f;
  for (int k = 0; k < nblk; ++k) s += slab[((long)plane * nblk + k) * 2 + which];
  means[plane * 2 + which] = s * inv_npos;
}

// ms = mean_planes prod_i v_i^w_i with v_i = relu(cs_i) (i < L - 1), relu(ssim_{L-1}); dmeans = d ms / d(those means)
__global__ void msssim_combine_kernel(const float* __restrict__ means, float* __restrict__ dmeans, float* __restrict__ ms_out, int levels,
                                      int planes, UirMsSsimWeights w) {
  __shared__ float red[256];
  float acc = 0.f;
  for (int p = threadIdx.x; p < planes; p += 256) {
    float v = 1.f;
    for (int i = 0; i < levels; ++i) {
      const float x = fmaxf(means[((long)i * planes + p) * 2 + (i == levels - 1 ? 0 : 1)], 0.f);
      v *= powf(x, w.w[i]);
    }
    acc += v;
    for (int i = 0; i < levels; ++i) {
      const int which = i == levels - 1 ? 0 : 1;
      const float x = means[((long)i * planes + p) * 2 + which];
      dmeans[((long)i * planes + p) * 2 + which] = x > 0.f ? v * w.w[i] / (x * (float)planes) : 0.f;
      dmeans[((long)i * planes + p) * 2 + (1 - which)] = 0.f;
    }
  }
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int s_ = 128; s_ > 0; s_ >>= 1) {
    if ((int)threadIdx.x < s_) red[threadIdx.x] += red[threadIdx.x + s_];
    __syncthreads();
  }
  if (threadIdx.x == 0) ms_out[0] = red[0] / (float)planes;
}

// The three derivative maps of one level at every valid position q, given gs = d L / d mean(ssim_map), gc = d L / d mean(cs_map)
// per plane:  with l = a / b, cs = c / d:  P = gs l + gc (on cs),  Q = gs cs (on l);
//   Bm = dL / d(G*x^2)[q] = P (-c / d^2),   Cm = dL / d(G*xy)[q] = P (2 / d),
//   A  = dL / d(mu1)[q]   = Q (2 mu2 b - 2 mu1 a) / b^2 - 2 mu1 Bm - mu2 Cm       (sigma terms contain -mu1^2, -mu1 mu2)
// so that  dL/dx[p] = sum_q g(p - q) A[q] + 2 x[p] sum_q g(p - q) Bm[q] + y[p] sum_q g(p - q) Cm[q]   (uir_msssim_adjoint).
__global__ __launch_bounds__(256) void msssim_bwd_maps_kernel(const UirMsSsimArgs a, const float* __restrict__ dmeans, float* __restrict__ A,
                                                              float* __restrict__ Bm, float* __restrict__ Cm) {
  const int plane = blockIdx.y;
  const int Ho = a.H - (WIN - 1), Wo = a.W - (WIN - 1);
  const long npos = (long)Ho * Wo;
  const float* X = a.X + (long)plane * a.H * a.W;
  const float* Y = a.Y + (long)plane * a.H * a.W;
  const float gs = dmeans[plane * 2] / (float)npos, gc = dmeans[plane * 2 + 1] / (float)npos;
  float g[WIN];
#pragma unroll
  for (int i = 0; i < WIN; ++i) g[i] = a.win[i];
  for (long p = (long)blockIdx.x * 256 + threadIdx.x; p < npos; p += (long)gridDim.x * 256) {
    const int oy = (int)(p / Wo), ox = (int)(p - (long)oy * Wo);
    const Moments m = moments_at(X, Y, a.W, oy, ox, g);
    const float s1 = m.xx - m.mu1 * m.mu1, s2 = m.yy - m.mu2 * m.mu2, s12 = m.xy - m.mu1 * m.mu2;
    const float c = 2.f * s12 + a.C2, d = s1 + s2 + a.C2;
    const float aa = 2.f * m.mu1 * m.mu2 + a.C1, b = m.mu1 * m.mu1 + m.mu2 * m.mu2 + a.C1;
    const float cs = c / d, l = aa / b;
    const float P = gs * l + gc, Q = gs * cs;
    const float bm = P * (-c / (d * d)), cm = P * (2.f / d);
    const long o = (long)plane * npos + p;
    Bm[o] = bm;
    Cm[o] = cm;
    A[o] = Q * (2.f * m.mu2 * b - 2.f * m.mu1 * aa) / (b * b) - 2.f * m.mu1 * bm - m.mu2 * cm;
  }
}

__global__ __launch_bounds__(256) void msssim_adjoint_kernel(const UirMsSsimArgs a, const float* __restrict__ A, const float* __restrict__ Bm,
                                                             const float* __restrict__ Cm, float* __restrict__ dX, int accumulate) {
  const int plane = blockIdx.y;
  const int Ho = a.H - (WIN - 1), Wo = a.W - (WIN - 1);
  const long npix = (long)a.H * a.W, npos = (long)Ho * Wo;
  const float* X = a.X + (long)plane * npix;
  const float* Y = a.Y + (long)plane * npix;
  const float* Ap = A + (long)plane * npos;
  const float* Bp = Bm + (long)plane * npos;
  const float* Cp = Cm + (long)plane * npos;
  float g[WIN];
#pragma unroll
  for (int i = 0; i < WIN; ++i) g[i] = a.win[i];
  for (long p = (long)blockIdx.x * 256 + threadIdx.x; p < npix; p += (long)gridDim.x * 256) {
    const int py = (int)(p / a.W), px = (int)(p - (long)py * a.W);
    float sa = 0.f, sb = 0.f, sc = 0.f;
    for (int dy = 0; dy < WIN; ++dy) {
      const int qy = py - dy;
      if (qy < 0 || qy >= Ho) continue;
      float ra = 0.f, rb = 0.f, rc = 0.f;
      for (int dx = 0; dx < WIN; ++dx) {
        const int qx = px - dx;
        if (qx < 0 || qx >= Wo) continue;
        const long q = (long)qy * Wo + qx;
        const float w = g[dx];
        ra = fmaf(w, Ap[q], ra); rb = fmaf(w, Bp[q], rb); rc = fmaf(w, Cp[q], rc);
      }
      const float wy = g[dy];
      sa = fmaf(wy, ra, sa); sb = fmaf(wy, rb, sb); sc = fmaf(wy, rc, sc);
    }
    const float v = sa + 2.f * X[p] * sb + Y[p] * sc;
    float* o = dX + (long)plane * npix + p;
    *o = accumulate ? *o + v : v;
  }
}

__global__ void avgpool2_kernel(const float* __restrict__ X, float* __restrict__ Yo, int H, int W, int ph, int pw, int Ho, int Wo, long n) {
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < n; idx += (long)gridDim.x * blockDim.x) {
    const long plane = idx / ((long)Ho * Wo);
    const int r = (int)(idx - plane * (long)Ho * Wo), oy = r / Wo, ox = r - oy * Wo;
    const float* xp = X + plane * (long)H * W;
    float s = 0.f;
#pragma unroll
    for (int dy = 0; dy < 2; ++dy)
#pragma unroll
      for (int dx = 0; dx < 2; ++dx) {
        const int y = 2 * oy - ph + dy, x = 2 * ox - pw + dx;
        if (y >= 0 && y < H && x >= 0 && x < W) s += xp[(long)y * W + x];
      }
    Yo[idx] = 0.25f * s;                                      // count_include_pad = True: always / 4
  }
}
__global__ void avgpool2_bwd_kernel(const float* __restrict__ dY, float* __restrict__ dX, int H, int W, int ph, int pw, int Ho, int Wo, long n,
                                    int accumulate) {
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < n; idx += (long)gridDim.x * blockDim.x) {
    const long plane = idx / ((long)H * W);
    const int r = (int)(idx - plane * (long)H * W), y = r / W, x = r - y * W;
    const int oy = (y + ph) >> 1, ox = (x + pw) >> 1;        // the one window that covers (y, x)
    const float v = (oy < Ho && ox < Wo) ? 0.25f * dY[plane * (long)Ho * Wo + (long)oy * Wo + ox] : 0.f;
    dX[idx] = accumulate ? dX[idx] + v : v;
  }
}
__global__ void axpy_dev_kernel(const float* __restrict__ x, const float* __restrict__ gscale, float coef, float* __restrict__ y, long n) {
  const float k = coef * gscale[0];
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < n; idx += (long)gridDim.x * blockDim.x) y[idx] = fmaf(k, x[idx], y[idx]);
}

unsigned grid_for(long n, int cap = 4096) {
  long g = (n + 255) / 256;
  return (unsigned)(g > cap ? cap : (g < 1 ? 1 : g));
}
int check(const UirMsSsimArgs& a) {
  UIR_REQUIRE(a.X && a.Y && a.planes > 0 && a.H >= WIN && a.W >= WIN && a.nblk > 0);
  return UIR_OK;
}

}  // namespace

extern "C" int uir_msssim_level(const UirMsSsimArgs* p, float* means, void* stream) {
  const UirMsSsimArgs& a = *p;
  if (int rc = check(a)) return rc;
  UIR_REQUIRE(a.slab && means);
  hipStream_t st = static_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(msssim_partials_kernel, dim3(a.nblk, a.planes), dim3(256), 0, st, a);
  const float inv = 1.0f / ((float)(a.H - WIN + 1) * (float)(a.W - WIN + 1));
  hipLaunchKernelGGL(msssim_means_kernel, dim3(a.planes), dim3(2), 0, st, a.slab, means, a.nblk, inv);
  return uir_launch_status();
}
extern "C" int uir_msssim_combine(const float* means, float* dmeans, float* ms_out, int levels, int planes, const UirMsSsimWeights* w,
                                  void* stream) {
  UIR_REQUIRE(means && dmeans && ms_out && w && levels >= 1 && levels <= 5 && planes > 0);
  hipLaunchKernelGGL(msssim_combine_kernel, dim3(1), dim3(256), 0, static_cast<hipStream_t>(stream), means, dmeans, ms_out, levels, planes, *w);
  return uir_launch_status();
}
extern "C" int uir_msssim_level_bwd(const UirMsSsimArgs* p, const float* dmeans, float* A, float* Bm, float* Cm, float* dX, int accumulate,
                                    void* stream) {
  const UirMsSsimArgs& a = *p;
  if (int rc = check(a)) return rc;
  UIR_REQUIRE(dmeans && A && Bm && Cm && dX);
  hipStream_t st = static_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(msssim_bwd_maps_kernel, dim3(a.nblk, a.planes), dim3(256), 0, st, a, dmeans, A, Bm, Cm);
  const long npix = (long)a.H * a.W;
  hipLaunchKernelGGL(msssim_adjoint_kernel, dim3(grid_for(npix, 1024), a.planes), dim3(256), 0, st, a, A, Bm, Cm, dX, accumulate);
  return uir_launch_status();
}
extern "C" int uir_avgpool2(const float* X, float* Y, int planes, int H, int W, void* stream) {
  UIR_REQUIRE(X && Y && planes > 0 && H > 0 && W > 0);
  const int ph = H & 1, pw = W & 1, Ho = (H + 2 * ph - 2) / 2 + 1, Wo = (W + 2 * pw - 2) / 2 + 1;
  const long n = (long)planes * Ho * Wo;
  hipLaunchKernelGGL(avgpool2_kernel, dim3(grid_for(n)), dim3(256), 0, static_cast<hipStream_t>(stream), X, Y, H, W, ph, pw, Ho, Wo, n);
  return uir_launch_status();
}
extern "C" int uir_avgpool2_bwd(const float* dY, float* dX, int planes, int H, int W, int accumulate, void* stream) {
  UIR_REQUIRE(dY && dX && planes > 0 && H > 0 && W > 0);
  const int ph = H & 1, pw = W & 1, Ho = (H + 2 * ph - 2) / 2 + 1, Wo = (W + 2 * pw - 2) / 2 + 1;
  const long n = (long)planes * H * W;
  hipLaunchKernelGGL(avgpool2_bwd_kernel, dim3(grid_for(n)), dim3(256), 0, static_cast<hipStream_t>(stream), dY, dX, H, W, ph, pw, Ho, Wo, n, accumulate);
  return uir_launch_status();
}
extern "C" int uir_axpy_dev(const float* x, const float* gscale, float coef, float* y, long n, void* stream) {
  UIR_REQUIRE(x && gscale && y && n > 0);
  hipLaunchKernelGGL(axpy_dev_kernel, dim3(grid_for(n)), dim3(256), 0, static_cast<hipStream_t>(stream), x, gscale, coef, y, n);
  return uir_launch_status();
}
