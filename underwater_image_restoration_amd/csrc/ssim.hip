// Validation SSIM (the part the trainer's validation loop uses): torchSSIM(ref, out) =
// pytorch_msssim.ssim(tar, prd, data_range=1.0, size_average=True) of src/ModelTrainer.py:16,23-24,149,156.
// The package is not vendored in the reference and absent from the image; this restates its published algorithm
// (v1.0: 11-tap Gaussian sigma 1.5 as a separable VALID convolution per channel of X, Y, X^2, Y^2, XY; C1 = (0.01 L)^2,
// C2 = (0.03 L)^2; ssim_map = (2 mu1 mu2 + C1) / (mu1^2 + mu2^2 + C1) * (2 s12 + C2) / (s1 + s2 + C2); mean over
// everything) -- PARITY UNPINNED against the package itself (no fixture of it exists here); oracle/ssim_ref.py is the
// torch-CPU restatement the tests compare with.
// One workgroup = one 16 x 32 tile of the valid output of one (sample, channel) plane: the 26 x 42 input tile of X and Y
// goes to LDS once, the five maps are blurred vertically into LDS, horizontally into registers, and the tile's sum of
// ssim_map values is one slab entry (summed in fixed order by uir_reduce_slab with scale = 1 / count).
#include "uir_common.h"
#include "../../include/uir_kernels.h"

namespace {

constexpr int TH = UIR_SSIM_TILE_H, TW = UIR_SSIM_TILE_W, WIN = 11, IH = TH + WIN - 1, IW = TW + WIN - 1;

__global__ __launch_bounds__(256) void uir_ssim_kernel(const UirSsimArgs a) {
  __shared__ float xs[IH][IW + 1], ys[IH][IW + 1];
  __shared__ float v[5][TH][IW + 1];
  __shared__ float red[4];
  const int tid = threadIdx.x;
  const int ox0 = blockIdx.x * TW, oy0 = blockIdx.y * TH;
  const long plane = (long)blockIdx.z * a.H * a.W;
  const float* __restrict__ X = a.X + plane;
  const float* __restrict__ Y = a.Y + plane;
  for (int e = tid; e < IH * IW; e += 256) {
    const int r = e / IW, c = e - r * IW;
    const int yy = min(oy0 + r, a.H - 1), xx = min(ox0 + c, a.W - 1);      // clamped: such taps only feed masked outputs
    xs[r][c] = X[(long)yy * a.W + xx];
    ys[r][c] = Y[(long)yy * a.W + xx];
  }
  __syncthreads();
  for (int e = tid; e < TH * IW; e += 256) {
    const int r = e / IW, c = e - r * IW;
    float sx = 0.f, sy = 0.f, sxx = 0.f, syy = 0.f, sxy = 0.f;
#pragma unroll
    for (int t = 0; t < WIN; ++t) {
      const float w = a.win[t], xv = xs[r + t][c], yv = ys[r + t][c];
      sx = fmaf(w, xv, sx);
      sy = fmaf(w, yv, sy);
      sxx = fmaf(w, xv * xv, sxx);
      syy = fmaf(w, yv * yv, syy);
      sxy = fmaf(w, xv * yv, sxy);
    }
    v[0][r][c] = sx; v[1][r][c] = sy; v[2][r][c] = sxx; v[3][r][c] = syy; v[4][r][c] = sxy;
  }
  __syncthreads();
  const int vh = a.H - WIN + 1, vw = a.W - WIN + 1;      // valid output extent
  float acc = 0.f;
  for (int e = tid; e < TH * TW; e += 256) {
    const int r = e / TW, c = e - r * TW;
    float m1 = 0.f, m2 = 0.f, e11 = 0.f, e22 = 0.f, e12 = 0.f;
#pragma unroll
    for (int t = 0; t < WIN; ++t) {
      const float w = a.win[t];
      m1 = fmaf(w, v[0][r][c + t], m1);
      m2 = fmaf(w, v[1][r][c + t], m2);
      e11 = fmaf(w, v[2][r][c + t], e11);
      e22 = fmaf(w, v[3][r][c + t], e22);
      e12 = fmaf(w, v[4][r][c + t], e12);
    }
    const float m11 = m1 * m1, m22 = m2 * m2, m12 = m1 * m2;
    const float s1 = e11 - m11, s2 = e22 - m22, s12 = e12 - m12;
    const float cs = (2.f * s12 + a.C2) / (s1 + s2 + a.C2);
    const float sv = ((2.f * m12 + a.C1) / (m11 + m22 + a.C1)) * cs;
    if (oy0 + r < vh && ox0 + c < vw) acc += sv;
  }
  acc = uir_wave_sum(acc);
  if ((tid & 63) == 0) red[tid >> 6] = acc;
  __syncthreads();
  if (tid == 0)
    a.slab[((long)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

}  // namespace

extern "C" long uir_ssim_nblocks(int planes, int H, int W) {
  if (planes < 1 || H < 11 || W < 11) return -1;
  const long gx = (W - 10 + UIR_SSIM_TILE_W - 1) / UIR_SSIM_TILE_W, gy = (H - 10 + UIR_SSIM_TILE_H - 1) / UIR_SSIM_TILE_H;
  return gx * gy * planes;
}

extern "C" int uir_ssim_partials(const UirSsimArgs* p, void* stream) {
  const UirSsimArgs& a = *p;
  UIR_REQUIRE(a.X && a.Y && a.slab && a.planes >= 1 && a.planes <= 65535);
  UIR_REQUIRE(a.H >= 11 && a.W >= 11);        // pytorch_msssim skips (with a warning) a dimension shorter than the window
  UIR_REQUIRE(a.slab_floats >= uir_ssim_nblocks(a.planes, a.H, a.W));
  const int gx = (a.W - 10 + TW - 1) / TW, gy = (a.H - 10 + TH - 1) / TH;
  UIR_REQUIRE(gy <= 65535);
  hipLaunchKernelGGL(uir_ssim_kernel, dim3(gx, gy, a.planes), dim3(256), 0, static_cast<hipStream_t>(stream), a);
  return uir_launch_status();
}
