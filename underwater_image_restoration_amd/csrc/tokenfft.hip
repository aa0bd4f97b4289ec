// Token-axis pieces of the NewBigFRFNModel EncoderBlock (reference src/model/model.py:57-92):
//   norm1(x)                                   nn.LayerNorm over C on (B, L, C) tokens  (:63)      uir_layernorm_fwd
//   fft.fftn(x, dim=(-2, -1)).real             a 2-D DFT over (L, C) - NOT a spatial fft2 (:72)    } Re(sum x e^{i(a+b)}) =
//   fft.ifftn(y, dim=(-2, -1)).real            the same operator / (L*C) on real input  (:86)       } C_L x C_C - S_L x S_C
// The C axis (32..128) is a pair of uir_gemm calls against cos / sin matrices; the L axis (L = H*W up to 65 536) is a
// two-stage Cooley-Tukey transform, L = N1*N2, on uir_dft_axis (MFMA contractions against host-built fp64 tables):
//   stage A over n1 (stride N2)  ->  uir_cmul_table: A[k1][n2] *= exp(2 pi i k1 n2 / L)  ->  stage B over n2  ->
//   uir_transpose_rows: frequency k = k1 + N1 k2 back to natural token order
// (three stages, L = N1*N2*N3, when that puts every stage on the two-stage MFMA DFT of csrc/spectral.hip: 65 536 = 64*32*32).
// The operator is real-symmetric, so its backward pass is the operator itself.
#include "uir_common.h"
#include "../../include/uir_kernels.h"

namespace {

template <int L>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const float* __restrict__ X, int ldx, const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, float* __restrict__ Y, int ldy,
                                                     float* __restrict__ stats, long M, float eps) {
  constexpr int ROWS = 256 / L;
  const int li = threadIdx.x % L, rl = threadIdx.x / L;
  const float4 g = *reinterpret_cast<const float4*>(gamma + 4 * li);
  const float4 b = *reinterpret_cast<const float4*>(beta + 4 * li);
  const float invc = 1.0f / (float)(4 * L);
  for (long m0 = (long)blockIdx.x * ROWS; m0 < M; m0 += (long)gridDim.x * ROWS) {
    const long m = m0 + rl;
    const bool ok = m < M;
    float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
    if (ok) x = *reinterpret_cast<const float4*>(X + m * ldx + 4 * li);
    float s = (x.x + x.y) + (x.z + x.w);
#pragma unroll
    for (int o = L / 2; o > 0; o >>= 1) s += __shfl_xor(s, o);
    const float mean = s * invc;
    const float d0 = x.x - mean, d1 = x.y - mean, d2 = x.z - mean, d3 = x.w - mean;
    float q = (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
#pragma unroll
    for (int o = L / 2; o > 0; o >>= 1) q += __shfl_xor(q, o);
    const float rstd = rsqrtf(q * invc + eps);
    if (ok) {
      float4 y;
      y.x = fmaf(d0 * rstd, g.x, b.x); y.y = fmaf(d1 * rstd, g.y, b.y); y.z = fmaf(d2 * rstd, g.z, b.z); y.w = fmaf(d3 * rstd, g.w, b.w);
      *reinterpret_cast<float4*>(Y + m * ldy + 4 * li) = y;
      if (li == 0 && stats) *reinterpret_cast<float2*>(stats + 2 * m) = make_float2(mean, rstd);
    }
  }
}

// C = 256 / 512 (the deep levels of NewBigFRFNModel): one wave per row, VPL float4 per lane
template <int VPL>
__global__ __launch_bounds__(256) void ln_fwd_wide_kernel(const float* __restrict__ X, int ldx, const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, float* __restrict__ Y, int ldy,
                                                          float* __restrict__ stats, long M, float eps) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float invc = 1.0f / (float)(256 * VPL);
  float4 g[VPL], b[VPL];
#pragma unroll
  for (int v = 0; v < VPL; ++v) {
    g[v] = *reinterpret_cast<const float4*>(gamma + 4 * (lane + 64 * v));
    b[v] = *reinterpret_cast<const float4*>(beta + 4 * (lane + 64 * v));
  }
  for (long m = (long)blockIdx.x * 4 + wave; m < M; m += (long)gridDim.x * 4) {
    float4 x[VPL];
    float s = 0.f;
#pragma unroll
    for (int v = 0; v < VPL; ++v) {
      x[v] = *reinterpret_cast<const float4*>(X + m * ldx + 4 * (lane + 64 * v));
      s += (x[v].x + x[v].y) + (x[v].z + x[v].w);
    }
    s = uir_wave_sum(s);
    const float mean = s * invc;
    float q = 0.f;
#pragma unroll
    for (int v = 0; v < VPL; ++v) {
      x[v].x -= mean; x[v].y -= mean; x[v].z -= mean; x[v].w -= mean;
      q += (x[v].x * x[v].x + x[v].y * x[v].y) + (x[v].z * x[v].z + x[v].w * x[v].w);
    }
    q = uir_wave_sum(q);
    const float rstd = rsqrtf(q * invc + eps);
#pragma unroll
    for (int v = 0; v < VPL; ++v) {
      float4 y;
      y.x = fmaf(x[v].x * rstd, g[v].x, b[v].x); y.y = fmaf(x[v].y * rstd, g[v].y, b[v].y);
      y.z = fmaf(x[v].z * rstd, g[v].z, b[v].z); y.w = fmaf(x[v].w * rstd, g[v].w, b[v].w);
      *reinterpret_cast<float4*>(Y + m * ldy + 4 * (lane + 64 * v)) = y;
    }
    if (lane == 0 && stats) *reinterpret_cast<float2*>(stats + 2 * m) = make_float2(mean, rstd);
  }
}

// (re, im)[b][p][c] *= (tr, ti)[p]  for p < P tokens, c < C channels (float4 along c)
__global__ __launch_bounds__(256) void cmul_table_kernel(float* __restrict__ re, float* __restrict__ im, const float* __restrict__ tr,
                                                         const float* __restrict__ ti, long rows, long P, int C4) {
  const long total = rows * C4;
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
    const long r = e / C4;
    const long p = r % P;
    const float wr = tr[p], wi = ti[p];
    float4 a = reinterpret_cast<float4*>(re)[e], b = reinterpret_cast<float4*>(im)[e];
    float4 oa, ob;
    oa.x = a.x * wr - b.x * wi; ob.x = a.x * wi + b.x * wr;
    oa.y = a.y * wr - b.y * wi; ob.y = a.y * wi + b.y * wr;
    oa.z = a.z * wr - b.z * wi; ob.z = a.z * wi + b.z * wr;
    oa.w = a.w * wr - b.w * wi; ob.w = a.w * wi + b.w * wr;
    reinterpret_cast<float4*>(re)[e] = oa;
    reinterpret_cast<float4*>(im)[e] = ob;
  }
}

// dst[b][k3][k2][k1][:] = scale * src[b][k1][k2][k3][:]   (rows of C floats; index reversal of a 2- or 3-stage transform:
// N2 = 1 gives the plain transpose dst[b][j][i] = src[b][i][j])
__global__ __launch_bounds__(256) void transpose_rows_kernel(const float* __restrict__ src, float* __restrict__ dst, int Bn, int N1,
                                                             int N2, int N3, int C4, float scale) {
  const long total = (long)Bn * N1 * N2 * N3 * C4;
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
    const int c = (int)(e % C4);
    long r = e / C4;                       // destination row: (b, k3, k2, k1)
    const int k1 = (int)(r % N1);
    r /= N1;
    const int k2 = (int)(r % N2);
    r /= N2;
    const int k3 = (int)(r % N3);
    const long b = r / N3;
    float4 v = reinterpret_cast<const float4*>(src)[(((b * N1 + k1) * N2 + k2) * N3 + k3) * C4 + c];
    v.x *= scale; v.y *= scale; v.z *= scale; v.w *= scale;
    reinterpret_cast<float4*>(dst)[e] = v;
  }
}

// Last stage of a length-N = 2*N2 DFT whose first stage ran over the odd / even sub-sequences (n = 2 n2 + n1):
//   Y rows [o][k2][n1][i][:]  ->  X[o][k1][k2][i][:] = scale * (Y[k2][0] + (-1)^k1 * tw[k2] * Y[k2][1]),  tw[k2] = exp(2 pi i k2 / N)
// i.e. the radix-2 butterfly with its twiddle, writing straight to natural frequency order (k = k1 N2 + k2).
__global__ __launch_bounds__(256) void radix2_combine_kernel(const float* __restrict__ yr, const float* __restrict__ yi,
                                                             float* __restrict__ xr, float* __restrict__ xi,
                                                             const float* __restrict__ tr, const float* __restrict__ ti, long O, int N2,
                                                             long IC4, float scale) {
  const long total = O * N2 * IC4;
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
    const long c = e % IC4;
    long r = e / IC4;
    const int k2 = (int)(r % N2);
    const long o = r / N2;
    const long src = ((o * N2 + k2) * 2) * IC4 + c;
    const float4 ar = reinterpret_cast<const float4*>(yr)[src], ai = reinterpret_cast<const float4*>(yi)[src];
    const float4 br = reinterpret_cast<const float4*>(yr)[src + IC4], bi = reinterpret_cast<const float4*>(yi)[src + IC4];
    const float wr = tr[k2], wi = ti[k2];
    float4 pr, pi;       // tw * b
    pr.x = br.x * wr - bi.x * wi; pi.x = br.x * wi + bi.x * wr;
    pr.y = br.y * wr - bi.y * wi; pi.y = br.y * wi + bi.y * wr;
    pr.z = br.z * wr - bi.z * wi; pi.z = br.z * wi + bi.z * wr;
    pr.w = br.w * wr - bi.w * wi; pi.w = br.w * wi + bi.w * wr;
    const long d0 = (o * 2 * N2 + k2) * IC4 + c, d1 = d0 + (long)N2 * IC4;
    reinterpret_cast<float4*>(xr)[d0] = make_float4((ar.x + pr.x) * scale, (ar.y + pr.y) * scale, (ar.z + pr.z) * scale, (ar.w + pr.w) * scale);
    reinterpret_cast<float4*>(xr)[d1] = make_float4((ar.x - pr.x) * scale, (ar.y - pr.y) * scale, (ar.z - pr.z) * scale, (ar.w - pr.w) * scale);
    if (xi) {
      reinterpret_cast<float4*>(xi)[d0] = make_float4((ai.x + pi.x) * scale, (ai.y + pi.y) * scale, (ai.z + pi.z) * scale, (ai.w + pi.w) * scale);
      reinterpret_cast<float4*>(xi)[d1] = make_float4((ai.x - pi.x) * scale, (ai.y - pi.y) * scale, (ai.z - pi.z) * scale, (ai.w - pi.w) * scale);
    }
  }
}

// dst[m][:] (+)= scale[m / rows_per_sample] * src[m][:]: stochastic depth (DropPath) - one factor per sample
__global__ __launch_bounds__(256) void scale_rows_kernel(const float* __restrict__ src, int lds, float* __restrict__ dst, int ldd,
                                                         const float* __restrict__ scale, long rows_per_sample, long M, int C4,
                                                         int accumulate) {
  const long total = M * C4;
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
    const long m = e / C4;
    const int c = (int)(e - m * C4) * 4;
    const float f = scale[m / rows_per_sample];
    float4 v = *reinterpret_cast<const float4*>(src + m * lds + c);
    v.x *= f; v.y *= f; v.z *= f; v.w *= f;
    if (accumulate) {
      const float4 o = *reinterpret_cast<const float4*>(dst + m * ldd + c);
      v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w;
    }
    *reinterpret_cast<float4*>(dst + m * ldd + c) = v;
  }
}

// dst[m][:] = a[m][:] + scale[m / rows_per_sample] * b[m][:]   (scale == nullptr: plain sum) - a residual join in ONE pass
__global__ __launch_bounds__(256) void add_scaled_rows_kernel(const float* __restrict__ a, int lda, const float* __restrict__ b, int ldb,
                                                              float* __restrict__ dst, int ldd, const float* __restrict__ scale,
                                                              long rows_per_sample, long M, int C4) {
  const long total = M * C4;
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
    const long m = e / C4;
    const int c = (int)(e - m * C4) * 4;
    const float f = scale ? scale[m / rows_per_sample] : 1.0f;
    const float4 x = *reinterpret_cast<const float4*>(a + m * lda + c);
    const float4 y = *reinterpret_cast<const float4*>(b + m * ldb + c);
    *reinterpret_cast<float4*>(dst + m * ldd + c) = make_float4(fmaf(f, y.x, x.x), fmaf(f, y.y, x.y), fmaf(f, y.z, x.z), fmaf(f, y.w, x.w));
  }
}

inline unsigned grid_for(long total) {
  long g = (total + 255) / 256;
  return (unsigned)(g > 8192 ? 8192 : (g < 1 ? 1 : g));
}

}  // namespace

extern "C" int uir_layernorm_fwd(const float* X, int ldx, const float* gamma, const float* beta, float* Y, int ldy, float* stats,
                                 long M, int C, float eps, void* stream) {
  UIR_REQUIRE(X && gamma && beta && Y && M > 0 && ldx >= C && ldy >= C && (ldx & 3) == 0 && (ldy & 3) == 0);
  UIR_REQUIRE(uir_aligned16(X) && uir_aligned16(Y) && uir_aligned16(gamma) && uir_aligned16(beta));
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int L = C / 4;
  long blocks = (M + (256 / (L > 0 ? L : 1)) - 1) / (256 / (L > 0 ? L : 1));
  if (blocks > 4096) blocks = 4096;
  switch (C) {
    case 16: hipLaunchKernelGGL(ln_fwd_kernel<4>, dim3((unsigned)blocks), dim3(256), 0, st, X, ldx, gamma, beta, Y, ldy, stats, M, eps); break;
    case 32: hipLaunchKernelGGL(ln_fwd_kernel<8>, dim3((unsigned)blocks), dim3(256), 0, st, X, ldx, gamma, beta, Y, ldy, stats, M, eps); break;
    case 64: hipLaunchKernelGGL(ln_fwd_kernel<16>, dim3((unsigned)blocks), dim3(256), 0, st, X, ldx, gamma, beta, Y, ldy, stats, M, eps); break;
    case 128: hipLaunchKernelGGL(ln_fwd_kernel<32>, dim3((unsigned)blocks), dim3(256), 0, st, X, ldx, gamma, beta, Y, ldy, stats, M, eps); break;
    case 256: hipLaunchKernelGGL(ln_fwd_wide_kernel<1>, dim3((unsigned)((M + 3) / 4 > 4096 ? 4096 : (M + 3) / 4)), dim3(256), 0, st, X, ldx, gamma, beta, Y, ldy, stats, M, eps); break;
    case 512: hipLaunchKernelGGL(ln_fwd_wide_kernel<2>, dim3((unsigned)((M + 3) / 4 > 4096 ? 4096 : (M + 3) / 4)), dim3(256), 0, st, X, ldx, gamma, beta, Y, ldy, stats, M, eps); break;
    default: return UIR_EINVAL;
  }
  return uir_launch_status();
}

extern "C" int uir_cmul_table(float* re, float* im, const float* tab_re, const float* tab_im, long rows, long P, int C, void* stream) {
  UIR_REQUIRE(re && im && tab_re && tab_im && rows > 0 && P > 0 && rows % P == 0 && C > 0 && (C & 3) == 0);
  UIR_REQUIRE(uir_aligned16(re) && uir_aligned16(im));
  hipLaunchKernelGGL(cmul_table_kernel, dim3(grid_for(rows * (C / 4))), dim3(256), 0, static_cast<hipStream_t>(stream), re, im, tab_re,
                     tab_im, rows, P, C / 4);
  return uir_launch_status();
}

extern "C" int uir_transpose_rows(const float* src, float* dst, int B, int N1, int N2, int N3, int C, float scale, void* stream) {
  UIR_REQUIRE(src && dst && src != dst && B > 0 && N1 > 0 && N2 > 0 && N3 > 0 && C > 0 && (C & 3) == 0);
  UIR_REQUIRE(uir_aligned16(src) && uir_aligned16(dst));
  hipLaunchKernelGGL(transpose_rows_kernel, dim3(grid_for((long)B * N1 * N2 * N3 * (C / 4))), dim3(256), 0,
                     static_cast<hipStream_t>(stream), src, dst, B, N1, N2, N3, C / 4, scale);
  return uir_launch_status();
}

extern "C" int uir_radix2_combine(const float* y_re, const float* y_im, float* x_re, float* x_im, const float* tab_re,
                                  const float* tab_im, long O, int N2, long I, int C, float scale, void* stream) {
  UIR_REQUIRE(y_re && y_im && x_re && tab_re && tab_im && O > 0 && N2 > 0 && I > 0 && C > 0 && (C & 3) == 0);
  UIR_REQUIRE(uir_aligned16(y_re) && uir_aligned16(y_im) && uir_aligned16(x_re) && (!x_im || uir_aligned16(x_im)));
  UIR_REQUIRE(x_re != y_re && x_re != y_im && x_im != y_re && x_im != y_im);
  const long IC4 = I * (C / 4);
  hipLaunchKernelGGL(radix2_combine_kernel, dim3(grid_for(O * N2 * IC4)), dim3(256), 0, static_cast<hipStream_t>(stream), y_re, y_im, x_re,
                     x_im, tab_re, tab_im, O, N2, IC4, scale);
  return uir_launch_status();
}

extern "C" int uir_add_scaled_rows(const float* a, int lda, const float* b, int ldb, float* dst, int ldd, const float* scale,
                                   long rows_per_sample, long M, int C, void* stream) {
  UIR_REQUIRE(a && b && dst && rows_per_sample > 0 && M > 0 && M % rows_per_sample == 0 && C > 0 && (C & 3) == 0);
  UIR_REQUIRE(lda >= C && ldb >= C && ldd >= C && ((lda | ldb | ldd) & 3) == 0 && uir_aligned16(a) && uir_aligned16(b) && uir_aligned16(dst));
  hipLaunchKernelGGL(add_scaled_rows_kernel, dim3(grid_for(M * (C / 4))), dim3(256), 0, static_cast<hipStream_t>(stream), a, lda, b, ldb,
                     dst, ldd, scale, rows_per_sample, M, C / 4);
  return uir_launch_status();
}

extern "C" int uir_scale_rows(const float* src, int lds, float* dst, int ldd, const float* scale, long rows_per_sample, long M, int C,
                              int accumulate, void* stream) {
  UIR_REQUIRE(src && dst && scale && rows_per_sample > 0 && M > 0 && M % rows_per_sample == 0 && C > 0 && (C & 3) == 0);
  UIR_REQUIRE(lds >= C && ldd >= C && ((lds | ldd) & 3) == 0 && uir_aligned16(src) && uir_aligned16(dst));
  hipLaunchKernelGGL(scale_rows_kernel, dim3(grid_for(M * (C / 4))), dim3(256), 0, static_cast<hipStream_t>(stream), src, lds, dst, ldd,
                     scale, rows_per_sample, M, C / 4, accumulate);
  return uir_launch_status();
}
