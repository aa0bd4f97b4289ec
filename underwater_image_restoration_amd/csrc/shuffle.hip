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

// 16-byte forms (C % 4 == 0, 16-byte rows, 32-bit index arithmetic): one thread per (input pixel, four channels) writes / reads float4.
// The scatter fills the 2 x 2 block of its pixel: the value on the even position, `fill` (or zeros) on the other three.
__global__ __launch_bounds__(256) void scatter_even4_kernel(const float* __restrict__ src, int lds, const float* __restrict__ fill,
                                                            float* __restrict__ dst, int ldd, int dcol, int H, int Wd, int C, unsigned n) {
  const unsigned cq_n = (unsigned)C >> 2;
  for (unsigned idx = blockIdx.x * 256u + threadIdx.x; idx < n; idx += gridDim.x * 256u) {      // idx over [B][H][W][C / 4]
    const unsigned cq = idx % cq_n, p = idx / cq_n;
    const unsigned w = p % (unsigned)Wd, q = p / (unsigned)Wd, h = q % (unsigned)H, b = q / (unsigned)H;
    const float4 v = *reinterpret_cast<const float4*>(src + (long)p * lds + 4 * cq);
    const float4 f = fill ? *reinterpret_cast<const float4*>(fill + 4 * cq) : make_float4(0.f, 0.f, 0.f, 0.f);
    const long r0 = ((long)(b * 2 * H + 2 * h) * 2 * Wd + 2 * w) * ldd + dcol + 4 * cq;
    const long dn = 2L * Wd * ldd;
    *reinterpret_cast<float4*>(dst + r0) = v;
    *reinterpret_cast<float4*>(dst + r0 + ldd) = f;
    *reinterpret_cast<float4*>(dst + r0 + dn) = f;
    *reinterpret_cast<float4*>(dst + r0 + dn + ldd) = f;
  }
}
__global__ __launch_bounds__(256) void gather_even4_kernel(const float* __restrict__ src, int lds, int scol, float* __restrict__ dst, int ldd,
                                                           int H, int Wd, int C, unsigned n) {
  const unsigned cq_n = (unsigned)C >> 2;
  for (unsigned idx = blockIdx.x * 256u + threadIdx.x; idx < n; idx += gridDim.x * 256u) {
    const unsigned cq = idx % cq_n, p = idx / cq_n;
    const unsigned w = p % (unsigned)Wd, q = p / (unsigned)Wd, h = q % (unsigned)H, b = q / (unsigned)H;
    *reinterpret_cast<float4*>(dst + (long)p * ldd + 4 * cq) =
        *reinterpret_cast<const float4*>(src + ((long)(b * 2 * H + 2 * h) * 2 * Wd + 2 * w) * lds + scol + 4 * cq);
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

// 16-byte form of the two kernels above (C % 4 == 0, 16-byte rows): one thread per (input pixel, four output channels) moves a 4 x 4
// block - four float4 in, transposed in registers, four float4 out - with 32-bit index arithmetic.  The element-wise forms read
// (shuffle) resp. write (unshuffle) 4-byte values 16 bytes apart and spend ~100 instructions per element on 64-bit divisions: 2.2 TB/s on
// ups_3's 2 x 67 MB, and the down-samplers' run on the chain.
__global__ __launch_bounds__(256) void pixel_shuffle4_kernel(const float* __restrict__ src, int lds, float* __restrict__ dst, int ldd, int dcol,
                                                             int H, int Wd, int C, unsigned n) {
  const unsigned cq_n = (unsigned)C >> 2;
  for (unsigned idx = blockIdx.x * 256u + threadIdx.x; idx < n; idx += gridDim.x * 256u) {      // idx over [B][H][W][C / 4]
    const unsigned cq = idx % cq_n, p = idx / cq_n;
    const unsigned w = p % (unsigned)Wd, q = p / (unsigned)Wd, h = q % (unsigned)H, b = q / (unsigned)H;
    const float4* s4 = reinterpret_cast<const float4*>(src + (long)p * lds + 16 * cq);
    const float4 v0 = s4[0], v1 = s4[1], v2 = s4[2], v3 = s4[3];                  // v_k = the four sub-pixels of channel 4 cq + k
    const long r0 = ((long)(b * 2 * H + 2 * h) * 2 * Wd + 2 * w) * ldd + dcol + 4 * cq;
    const long dn = 2L * Wd * ldd;
    *reinterpret_cast<float4*>(dst + r0) = make_float4(v0.x, v1.x, v2.x, v3.x);
    *reinterpret_cast<float4*>(dst + r0 + ldd) = make_float4(v0.y, v1.y, v2.y, v3.y);
    *reinterpret_cast<float4*>(dst + r0 + dn) = make_float4(v0.z, v1.z, v2.z, v3.z);
    *reinterpret_cast<float4*>(dst + r0 + dn + ldd) = make_float4(v0.w, v1.w, v2.w, v3.w);
  }
}
__global__ __launch_bounds__(256) void pixel_unshuffle4_kernel(const float* __restrict__ src, int lds, int scol, float* __restrict__ dst, int ldd,
                                                               int H, int Wd, int C, unsigned n) {
  const unsigned cq_n = (unsigned)C >> 2;
  for (unsigned idx = blockIdx.x * 256u + threadIdx.x; idx < n; idx += gridDim.x * 256u) {      // idx over [B][H][W][C / 4]
    const unsigned cq = idx % cq_n, p = idx / cq_n;
    const unsigned w = p % (unsigned)Wd, q = p / (unsigned)Wd, h = q % (unsigned)H, b = q / (unsigned)H;
    const long r0 = ((long)(b * 2 * H + 2 * h) * 2 * Wd + 2 * w) * lds + scol + 4 * cq;
    const long dn = 2L * Wd * lds;
    const float4 u0 = *reinterpret_cast<const float4*>(src + r0), u1 = *reinterpret_cast<const float4*>(src + r0 + lds);
    const float4 u2 = *reinterpret_cast<const float4*>(src + r0 + dn), u3 = *reinterpret_cast<const float4*>(src + r0 + dn + lds);
    float4* d4 = reinterpret_cast<float4*>(dst + (long)p * ldd + 16 * cq);
    d4[0] = make_float4(u0.x, u1.x, u2.x, u3.x);
    d4[1] = make_float4(u0.y, u1.y, u2.y, u3.y);
    d4[2] = make_float4(u0.z, u1.z, u2.z, u3.z);
    d4[3] = make_float4(u0.w, u1.w, u2.w, u3.w);
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

// Patches of a 4 x 4, stride 2, padding 1 convolution (Conv2d of Downsample, src/Models/AST.py:408-424) in the weight's OWN k order:
// P[(b, oy, ox)][c * 16 + ky * 4 + kx] = x[(b, 2 oy - 1 + ky, 2 ox - 1 + kx)][c] (0 outside the image), so the convolution is one
// dense contraction against the native [Cout][Cin * 16] weight.  One thread per (output pixel, channel): 16 reads coalesced over the
// channels of a pixel row, 64 contiguous bytes written.  H, Wd are the OUTPUT sides.
__global__ void patch4s2_kernel(const float* __restrict__ x, int ldx, float* __restrict__ P, int ldp, int H, int Wd, int C, long n) {
  GRID_STRIDE(idx, n) {  // idx over [B][H][W][C]
    const int c = (int)(idx % C);
    const long pix = idx / C;
    const int ox = (int)(pix % Wd);
    const long q = pix / Wd;
    const int oy = (int)(q % H);
    const long b = q / H;
    float4 v[4];
#pragma unroll
    for (int ky = 0; ky < 4; ++ky) {
      const int iy = 2 * oy - 1 + ky;
      const bool yok = iy >= 0 && iy < 2 * H;
      const float* row = x + ((b * 2 * H + (yok ? iy : 0)) * 2 * Wd) * (long)ldx + c;
      float t[4];
#pragma unroll
      for (int kx = 0; kx < 4; ++kx) {
        const int ix = 2 * ox - 1 + kx;
        const bool ok = yok && ix >= 0 && ix < 2 * Wd;
        const float u = row[(long)(ok ? ix : 0) * ldx];
        t[kx] = ok ? u : 0.f;
      }
      v[ky] = make_float4(t[0], t[1], t[2], t[3]);
    }
    float4* d = reinterpret_cast<float4*>(P + pix * ldp + 16 * c);
    d[0] = v[0]; d[1] = v[1]; d[2] = v[2]; d[3] = v[3];
  }
}

// ... and its adjoint: dx[(b, iy, ix)][c] = sum of the (at most four) patch entries that hold this pixel, in (ky, kx) order.
// H, Wd are the sides of the PATCH grid (dx has 2 H x 2 Wd pixels).
__global__ void patch4s2_bwd_kernel(const float* __restrict__ dP, int ldp, float* __restrict__ dx, int ldx, int H, int Wd, int C, long n) {
  GRID_STRIDE(idx, n) {  // idx over [B][2H][2W][C]
    const int c = (int)(idx % C);
    const long pix = idx / C;
    const int ix = (int)(pix % (2 * Wd));
    const long q = pix / (2 * Wd);
    const int iy = (int)(q % (2 * H));
    const long b = q / (2 * H);
    float s = 0.f;
#pragma unroll
    for (int a = 0; a < 2; ++a) {
      const int ky = ((iy + 1) & 1) + 2 * a, oy = (iy + 1 - ky) >> 1;      // iy = 2 oy - 1 + ky
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const int kx = ((ix + 1) & 1) + 2 * e, ox = (ix + 1 - kx) >> 1;
        const bool ok = oy >= 0 && oy < H && ox >= 0 && ox < Wd;
        const float u = dP[((b * H + (ok ? oy : 0)) * Wd + (ok ? ox : 0)) * (long)ldp + 16 * c + 4 * ky + kx];
        s += ok ? u : 0.f;
      }
    }
    dx[pix * ldx + c] = s;
  }
}

// Patches of a 3 x 3, padding 1 convolution in the weight's own k order: P[(b, y, x)][c * 9 + ky * 3 + kx] = x[(b, y - 1 + ky,
// x - 1 + kx)][c] (0 outside), and the adjoint.  For the WIDE convolutions on FEW pixels (samplers of the deep U-Net levels of
// NewBigFRFNModel, block.py:107-153; VGG16's conv3 / conv4 stacks on a batch of 4): 128 x 128 output tiles of the in-place nine-tap
// kernel leave most of the chip idle there (2 048 pixels x 1 024 channels = 128 workgroups, flipped-tap weights read 4 bytes at a
// time), as patches the layer is one plain K = 9 Cin contraction with split-k (csrc/tiled.hip) and the patch matrix is a few MB.
__global__ void patch3x3_kernel(const float* __restrict__ x, int ldx, float* __restrict__ P, int ldp, int H, int Wd, int C, long n) {
  GRID_STRIDE(idx, n) {  // idx over [B][H][W][C]
    const int c = (int)(idx % C);
    const long pix = idx / C;
    const int px = (int)(pix % Wd);
    const int py = (int)((pix / Wd) % H);
    float* d = P + pix * ldp + 9 * c;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const int dy = t / 3 - 1, dx = t % 3 - 1;
      const bool ok = py + dy >= 0 && py + dy < H && px + dx >= 0 && px + dx < Wd;
      const float u = x[(pix + (ok ? (long)dy * Wd + dx : 0)) * ldx + c];
      d[t] = ok ? u : 0.f;
    }
  }
}

__global__ void patch3x3_bwd_kernel(const float* __restrict__ dP, int ldp, float* __restrict__ dx, int ldx, int H, int Wd, int C, long n) {
  GRID_STRIDE(idx, n) {  // idx over [B][H][W][C]
    const int c = (int)(idx % C);
    const long pix = idx / C;
    const int px = (int)(pix % Wd);
    const int py = (int)((pix / Wd) % H);
    float s = 0.f;
#pragma unroll
    for (int t = 0; t < 9; ++t) {                // patch (y - dy, x - dx) holds this pixel at tap t
      const int dy = t / 3 - 1, dxx = t % 3 - 1;
      const bool ok = py - dy >= 0 && py - dy < H && px - dxx >= 0 && px - dxx < Wd;
      const float u = dP[(pix - (ok ? (long)dy * Wd + dxx : 0)) * ldp + 9 * c + t];
      s += ok ? u : 0.f;
    }
    dx[pix * ldx + c] = s;
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

// flat device-to-device copy as an ordinary kernel (float4 per thread): the module networks copy each module's input / output gradient into
// its plan's buffer - as hipMemcpyAsync (torch's copy_) each of those ~54 copies per step left the queue idle for 30-60 us around the blit
__global__ void copy_flat4_kernel(const float4* __restrict__ src, float4* __restrict__ dst, long n4) {
  GRID_STRIDE(idx, n4) dst[idx] = src[idx];
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
  if ((C & 3) == 0 && (lds & 3) == 0 && (ldd & 3) == 0 && (dcol & 3) == 0 && uir_aligned16(src) && uir_aligned16(dst) && (!fill || uir_aligned16(fill)) &&
      n / 16 < (1L << 31)) {
    const unsigned n4 = (unsigned)(n / 16);
    const unsigned blocks = (n4 + 255) / 256 < 65536 ? (n4 + 255) / 256 : 65536;
    hipLaunchKernelGGL(scatter_even4_kernel, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream), src, lds, fill, dst, ldd, dcol, H, Wd, C, n4);
    return uir_launch_status();
  }
  hipLaunchKernelGGL(scatter_even_kernel, ew_grid(n), dim3(256), 0, static_cast<hipStream_t>(stream), src, lds, fill, dst, ldd, dcol, H, Wd, C, n);
  return uir_launch_status();
}
extern "C" int uir_gather_even(const float* src, int lds, int scol, float* dst, int ldd, int B, int H, int Wd, int C, void* stream) {
  UIR_REQUIRE(src && dst && B > 0 && H > 0 && Wd > 0 && C > 0 && lds >= scol + C && ldd >= C);
  const long n = (long)B * H * Wd * C;
  if ((C & 3) == 0 && (lds & 3) == 0 && (ldd & 3) == 0 && (scol & 3) == 0 && uir_aligned16(src) && uir_aligned16(dst) && n / 4 < (1L << 31)) {
    const unsigned n4 = (unsigned)(n / 4);
    const unsigned blocks = (n4 + 255) / 256 < 65536 ? (n4 + 255) / 256 : 65536;
    hipLaunchKernelGGL(gather_even4_kernel, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream), src, lds, scol, dst, ldd, H, Wd, C, n4);
    return uir_launch_status();
  }
  hipLaunchKernelGGL(gather_even_kernel, ew_grid(n), dim3(256), 0, static_cast<hipStream_t>(stream), src, lds, scol, dst, ldd, H, Wd, C, n);
  return uir_launch_status();
}
extern "C" int uir_pixel_shuffle(const float* src, int lds, float* dst, int ldd, int dcol, int B, int H, int Wd, int C, void* stream) {
  UIR_REQUIRE(src && dst && B > 0 && H > 0 && Wd > 0 && C > 0 && lds >= 4 * C && ldd >= dcol + C);
  const long n = (long)B * 4 * H * Wd * C;
  if ((C & 3) == 0 && (lds & 3) == 0 && (ldd & 3) == 0 && (dcol & 3) == 0 && uir_aligned16(src) && uir_aligned16(dst) && n / 16 < (1L << 31) &&
      (long)B * H * Wd * (long)lds < (1L << 40)) {
    const unsigned n4 = (unsigned)(n / 16);
    const unsigned blocks = (n4 + 255) / 256 < 65536 ? (n4 + 255) / 256 : 65536;
    hipLaunchKernelGGL(pixel_shuffle4_kernel, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream), src, lds, dst, ldd, dcol, H, Wd, C, n4);
    return uir_launch_status();
  }
  hipLaunchKernelGGL(pixel_shuffle_kernel, ew_grid(n), dim3(256), 0, static_cast<hipStream_t>(stream), src, lds, dst, ldd, dcol, H, Wd, C, n);
  return uir_launch_status();
}
extern "C" int uir_pixel_unshuffle(const float* src, int lds, int scol, float* dst, int ldd, int B, int H, int Wd, int C, void* stream) {
  UIR_REQUIRE(src && dst && B > 0 && H > 0 && Wd > 0 && C > 0 && lds >= scol + C && ldd >= 4 * C);
  const long n = (long)B * H * Wd * 4 * C;
  if ((C & 3) == 0 && (lds & 3) == 0 && (ldd & 3) == 0 && (scol & 3) == 0 && uir_aligned16(src) && uir_aligned16(dst) && n / 16 < (1L << 31)) {
    const unsigned n4 = (unsigned)(n / 16);
    const unsigned blocks = (n4 + 255) / 256 < 65536 ? (n4 + 255) / 256 : 65536;
    hipLaunchKernelGGL(pixel_unshuffle4_kernel, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream), src, lds, scol, dst, ldd, H, Wd, C, n4);
    return uir_launch_status();
  }
  hipLaunchKernelGGL(pixel_unshuffle_kernel, ew_grid(n), dim3(256), 0, static_cast<hipStream_t>(stream), src, lds, scol, dst, ldd, H, Wd, C, n);
  return uir_launch_status();
}
extern "C" int uir_patch4s2(const float* x, int ldx, float* P, int ldp, int B, int H, int Wd, int C, void* stream) {
  UIR_REQUIRE(x && P && B > 0 && H > 0 && Wd > 0 && C > 0 && ldx >= C && ldp >= 16 * C && (ldp & 3) == 0 && uir_aligned16(P));
  const long n = (long)B * H * Wd * C;
  hipLaunchKernelGGL(patch4s2_kernel, ew_grid(n), dim3(256), 0, static_cast<hipStream_t>(stream), x, ldx, P, ldp, H, Wd, C, n);
  return uir_launch_status();
}
extern "C" int uir_patch4s2_bwd(const float* dP, int ldp, float* dx, int ldx, int B, int H, int Wd, int C, void* stream) {
  UIR_REQUIRE(dP && dx && B > 0 && H > 0 && Wd > 0 && C > 0 && ldx >= C && ldp >= 16 * C);
  const long n = (long)B * H * Wd * 4 * C;
  hipLaunchKernelGGL(patch4s2_bwd_kernel, ew_grid(n), dim3(256), 0, static_cast<hipStream_t>(stream), dP, ldp, dx, ldx, H, Wd, C, n);
  return uir_launch_status();
}
extern "C" int uir_patch3x3(const float* x, int ldx, float* P, int ldp, int B, int H, int Wd, int C, void* stream) {
  UIR_REQUIRE(x && P && B > 0 && H > 0 && Wd > 0 && C > 0 && ldx >= C && ldp >= 9 * C);
  const long n = (long)B * H * Wd * C;
  hipLaunchKernelGGL(patch3x3_kernel, ew_grid(n), dim3(256), 0, static_cast<hipStream_t>(stream), x, ldx, P, ldp, H, Wd, C, n);
  return uir_launch_status();
}
extern "C" int uir_patch3x3_bwd(const float* dP, int ldp, float* dx, int ldx, int B, int H, int Wd, int C, void* stream) {
  UIR_REQUIRE(dP && dx && B > 0 && H > 0 && Wd > 0 && C > 0 && ldx >= C && ldp >= 9 * C);
  const long n = (long)B * H * Wd * C;
  hipLaunchKernelGGL(patch3x3_bwd_kernel, ew_grid(n), dim3(256), 0, static_cast<hipStream_t>(stream), dP, ldp, dx, ldx, H, Wd, C, n);
  return uir_launch_status();
}
extern "C" int uir_copy_cols(const float* src, int lds, int scol, float* dst, int ldd, int dcol, long M, int C, int accumulate, void* stream) {
  UIR_REQUIRE(src && dst && M > 0 && C > 0 && lds >= scol + C && ldd >= dcol + C);
  const long n = M * C;
  hipLaunchKernelGGL(copy_cols_kernel, ew_grid(n), dim3(256), 0, static_cast<hipStream_t>(stream), src, lds, scol, dst, ldd, dcol, C, accumulate, n);
  return uir_launch_status();
}
extern "C" int uir_copy_flat(const float* src, float* dst, long n, void* stream) {
  UIR_REQUIRE(src && dst && n > 0 && (n & 3) == 0 && uir_aligned16(src) && uir_aligned16(dst));
  hipLaunchKernelGGL(copy_flat4_kernel, ew_grid(n / 4), dim3(256), 0, static_cast<hipStream_t>(stream),
                     reinterpret_cast<const float4*>(src), reinterpret_cast<float4*>(dst), n / 4);
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
