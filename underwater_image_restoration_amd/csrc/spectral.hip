// Spectral up-sampler pieces (reference UpSample, SpectralTransformer.py:174-188):
//  * uir_dft_axis: one axis of the 2-D DFT of an HWC tensor as an f32-MFMA contraction with
//    cos/sin twiddle tables - any length N, channels on the MFMA lane so every operand load and
//    result store is a 128-B contiguous HWC channel run, no bit reversal, no LDS.
//  * magnitude/phase, polar -> complex, |z| and their adjoints (element-wise, HBM-bound, tiny).
// The reference's ifft2 of the 2x2-tiled spectrum on 2H x 2W equals ifft2 on H x W written to the
// even grid (zeros elsewhere), so only H x W transforms are ever computed here.
#include "uir_common.h"
#include "../../include/uir_kernels.h"

namespace {

constexpr int KT = 2;  // 32-frequency tiles per wave

__global__ __launch_bounds__(256) void uir_dft_kernel(const UirDftArgs a, int ctiles, int kgroups, long items) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 31, h2 = lane >> 5;
  const long item = (long)blockIdx.x * 4 + wave;
  if (item >= items) return;
  const int kg = (int)(item % kgroups);
  long t = item / kgroups;
  const int ct = (int)(t % ctiles);
  t /= ctiles;
  const int i = (int)(t % a.I);
  const long o = t / a.I;
  const int c = ct * 32 + r;
  const bool cok = c < a.C;
  const int k0 = kg * 32 * KT;
  const bool has_im = a.in_im != nullptr;
  const bool want_im = a.out_im != nullptr;

  f32x16 are[KT], aim[KT];
#pragma unroll
  for (int kt = 0; kt < KT; ++kt)
#pragma unroll
    for (int q = 0; q < 16; ++q) { are[kt][q] = 0.f; aim[kt][q] = 0.f; }

  const long line = (o * a.N * a.I + i) * a.C;  // element (o, n=0, i, c=0)
  const long nstride = (long)a.I * a.C;
  // branch-free loads (clamped indices + select), U n-pairs per trip so 2U..4U loads are in flight
  const int cc = min(c, a.C - 1);
  const float* imp = has_im ? a.in_im : a.in_re;   // always a valid address; zeroed by select when real-only
  int kcl[KT];
  bool kok[KT];
#pragma unroll
  for (int kt = 0; kt < KT; ++kt) { const int k = k0 + 32 * kt + r; kok[kt] = k < a.N; kcl[kt] = min(k, a.N - 1); }
  constexpr int U = 4;
  for (int n = 0; n < a.N; n += 2 * U) {
    float bre[U], bim[U], cv[U][KT], sv[U][KT];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int nn = min(n + 2 * u + h2, a.N - 1);
      bre[u] = a.in_re[line + nn * nstride + cc];
      bim[u] = imp[line + nn * nstride + cc];
#pragma unroll
      for (int kt = 0; kt < KT; ++kt) {
        cv[u][kt] = a.cos_t[(long)nn * a.N + kcl[kt]];
        sv[u][kt] = a.sin_t[(long)nn * a.N + kcl[kt]];
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const bool nok = (n + 2 * u + h2) < a.N;
      const float br = (nok && cok) ? bre[u] : 0.f;
      const float bi = (nok && cok && has_im) ? bim[u] : 0.f;
#pragma unroll
      for (int kt = 0; kt < KT; ++kt) {
        const float c_ = (nok && kok[kt]) ? cv[u][kt] : 0.f;
        const float s_ = (nok && kok[kt]) ? a.sign * sv[u][kt] : 0.f;
        // e^{i s th}(re + i im) = (cos re - s sin im) + i (cos im + s sin re)
        are[kt] = uir_mfma(c_, br, are[kt]);
        if (has_im) are[kt] = uir_mfma(-s_, bi, are[kt]);
        if (want_im) {
          aim[kt] = uir_mfma(s_, br, aim[kt]);
          if (has_im) aim[kt] = uir_mfma(c_, bi, aim[kt]);
        }
      }
    }
  }
  if (!cok) return;
#pragma unroll
  for (int kt = 0; kt < KT; ++kt)
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int k = k0 + 32 * kt + uir_mfma_row(q, h2);
      if (k < a.N) {
        a.out_re[line + k * nstride + c] = are[kt][q] * a.scale;
        if (want_im) a.out_im[line + k * nstride + c] = aim[kt][q] * a.scale;
      }
    }
}

// Two-stage (Cooley-Tukey 16 x N2) DFT axis for N = 16 * N2 in {32, 64, 128}: 4x fewer MFMAs than the dense
// contraction at N = 128.  n = N2 n1 + n2, k = k1 + 16 k2:
//   stage A (per n2):  Y[n2][k1] = sum_n1 W16^(n1 k1) x[N2 n1 + n2]      one 32x32 real tile: rows (Re|Im, k1), K = (re|im, n1)
//   stage B (per k1):  X[k1 + 16 k2] = sum_n2 W^(n2 (k1 + 16 k2)) Y[n2][k1]   (the twiddle W^(n2 k1) is folded into the matrix)
// Stage A leaves Y in the MFMA accumulator layout (lane = channel, register = (Re|Im, k1)): lane half 0 owns
// k1 in {0-3, 8-11}, half 1 owns k1 + 4.  Stage B therefore transforms the PAIR (k1, k1 + 4) per MFMA tile with a
// block-diagonal operand (rows (which k1, Re|Im, k2), contraction (n2, Re|Im) x lane half): its B operand is simply
// accumulator register j (+8 for Im) of stage A - no shuffles, no LDS.  Operand tables are built on the host in fp64.
// PRO / EPI (UirDft2Args.pro / .epi): the element-wise op in front of / behind the transform is applied where its operand sits in
// registers - on the loaded line resp. on the (Re, Im) register pair of an output element - with the arithmetic of the stand-alone
// kernels below (uir_polar_fwd, uir_cabs_bwd, uir_magpha_bwd / uir_magpha_fwd, uir_cabs_fwd, uir_polar_bwd).
template <int N2, int PRO, int EPI>
__global__ __launch_bounds__(256) void uir_dft2_kernel(const UirDft2Args a, int ctiles, long items) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int li = lane & 31, h2 = lane >> 5;
  const long item = (long)blockIdx.x * 4 + wave;
  if (item >= items) return;
  const int ct = (int)(item % ctiles);
  long t = item / ctiles;
  const int i = (int)(t % a.I);
  const long o = t / a.I;
  constexpr int N = 16 * N2;
  const int c = ct * 32 + li;
  const bool cok = c < a.C;
  const int cc = min(c, a.C - 1);
  const long nstride = (long)a.I * a.C;
  const long line = (o * N * a.I + i) * a.C;
  const bool has_im = a.in_im != nullptr, want_im = a.out_im != nullptr;
  const float* imp = has_im ? a.in_im : a.in_re;

  float ta[16];
#pragma unroll
  for (int s = 0; s < 16; ++s) ta[s] = a.ta[s * 64 + lane];

  f32x16 D[N2];
  // fused-path trigonometry: hardware sine / cosine behind a fract range reduction (__sincosf, ~1e-6 absolute); the library sincosf's
  // large-argument path needs a scratch table, which costs this register-full kernel its occupancy
  auto conv = [&](float& re, float& im, float x0, float x1, float x2) {
    if (PRO == UIR_DFT_PRO_POLAR) {
      float sn, cs;
      __sincosf(im, &sn, &cs);
      const float m = re;
      re = m * cs;
      im = m * sn;
    } else if (PRO == UIR_DFT_PRO_CABS_BWD) {
      const float q = x2 > 0.f ? re / x2 : 0.f;
      re = q * x0;
      im = q * x1;
    } else if (PRO == UIR_DFT_PRO_MAGPHA_BWD) {
      float sn, cs;
      __sincosf(x1, &sn, &cs);
      float ga = 0.f, gb = 0.f;
      if (x0 > 0.f) {
        const float q = im / x0;
        ga = re * cs - q * sn;
        gb = re * sn + q * cs;
      }
      re = ga;
      im = gb;
    }
  };
#pragma unroll
  for (int n2 = 0; n2 < N2; ++n2) {
    f32x16 d;
#pragma unroll
    for (int q = 0; q < 16; ++q) d[q] = 0.f;
    // PRO: two half batches of 4 k-steps (their extra operands would not fit beside the N2 accumulators otherwise)
    constexpr int NB = PRO == UIR_DFT_PRO_NONE ? 1 : 2, SB = 8 / NB;
#pragma unroll
    for (int hb = 0; hb < NB; ++hb) {
      float br[SB], bi[SB], x0[SB], x1[SB], x2[SB];
#pragma unroll
      for (int t = 0; t < SB; ++t) {
        const int s = SB * hb + t;
        const long off = line + (long)(N2 * (2 * s + h2) + n2) * nstride + cc;
        br[t] = a.in_re[off];
        bi[t] = imp[off];
        x0[t] = x1[t] = x2[t] = 0.f;
        if (PRO == UIR_DFT_PRO_CABS_BWD) { x0[t] = a.aux0[off]; x1[t] = a.aux1[off]; x2[t] = a.aux2[off]; }
        if (PRO == UIR_DFT_PRO_MAGPHA_BWD) { x0[t] = a.aux0[off]; x1[t] = a.aux1[off]; }
      }
      // all loads of the batch in flight before its MFMAs: hipcc sinks each imaginary-part load next to the MFMA that consumes it
      // (`global_load_dword; s_waitcnt vmcnt(0); v_mfma`, eight serialised round trips per n2) when left to schedule freely
      __builtin_amdgcn_sched_barrier(0);
      if (PRO != UIR_DFT_PRO_NONE) {
#pragma unroll
        for (int t = 0; t < SB; ++t) conv(br[t], bi[t], x0[t], x1[t], x2[t]);
      }
#pragma unroll
      for (int t = 0; t < SB; ++t) d = uir_mfma(ta[SB * hb + t], cok ? br[t] : 0.f, d);
      if (has_im || PRO != UIR_DFT_PRO_NONE) {
#pragma unroll
        for (int t = 0; t < SB; ++t) d = uir_mfma(ta[8 + SB * hb + t], cok ? bi[t] : 0.f, d);
      }
    }
    D[n2] = d;
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    float tb[2 * N2];
#pragma unroll
    for (int s = 0; s < 2 * N2; ++s) tb[s] = a.tb[(j * 2 * N2 + s) * 64 + lane];
    f32x16 e;
#pragma unroll
    for (int q = 0; q < 16; ++q) e[q] = 0.f;
#pragma unroll
    for (int s = 0; s < 2 * N2; ++s) e = uir_mfma(tb[s], D[s >> 1][j + 8 * (s & 1)], e);
    const int aj = (j & 3) + 8 * (j >> 2);          // k1 of lane half 0; half 1 holds k1 + 4 ... as ROW blocks w = 0 / 1
    if (cok && EPI == UIR_DFT_EPI_NONE) {
      // plain transform: aux0 (optional, real output only) is ADDED to the result - the up-sampler's spectral input gradient lands on the
      // shuffle branch's without a separate accumulate pass (it may alias out_re: every element is read and written by this lane alone)
      const float* add = (PRO == UIR_DFT_PRO_NONE && !want_im) ? a.aux0 : nullptr;
      float av[16];                                  // every addend first: the loads cannot be moved across the (possibly aliasing) stores by the compiler
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int k2 = (q & 3) + 4 * h2, p = (q >> 2) & 1, w = q >> 3;
        av[q] = (add && k2 < N2 && p == 0) ? add[line + (long)(aj + 4 * w + 16 * k2) * nstride + c] : 0.f;
      }
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int k2 = (q & 3) + 4 * h2, p = (q >> 2) & 1, w = q >> 3;
        if (k2 < N2 && (p == 0 || want_im)) {
          const int k = aj + 4 * w + 16 * k2;
          float* dst = p ? a.out_im : a.out_re;
          const long at = line + (long)k * nstride + c;
          dst[at] = fmaf(e[q], a.scale, av[q]);
        }
      }
    }
    if (cok && EPI != UIR_DFT_EPI_NONE) {
      // output element k: Re in register q (p = 0), Im in register q + 4
#pragma unroll
      for (int qq = 0; qq < 8; ++qq) {
        const int q = (qq & 3) + 8 * (qq >> 2);
        const int k2 = (q & 3) + 4 * h2, w = q >> 3;
        if (k2 < N2) {
          const int k = aj + 4 * w + 16 * k2;
          const long dst = line + (long)k * nstride + c;
          const float re = e[q] * a.scale;
          float im = e[q + 4] * a.scale;
          if (EPI == UIR_DFT_EPI_MAGPHA) {
            // H-axis transform of [B][H][W][C]: h = k, w = i; Im is exactly +0 at the self-conjugate bins of a real signal
            const bool hs = (k == 0) || (((N & 1) == 0) && k == N / 2);
            const bool ws = (i == 0) || (((a.I & 1) == 0) && i == a.I / 2);
            if (hs && ws) im = 0.0f;
            a.out_re[dst] = sqrtf(re * re + im * im);
            a.out_im[dst] = atan2f(im, re);
          } else if (EPI == UIR_DFT_EPI_CABS) {
            a.out_re[dst] = re;
            a.out_im[dst] = im;
            a.aux_out[dst] = sqrtf(re * re + im * im);
          } else {
            float sn, cs;
            __sincosf(a.aux1[dst], &sn, &cs);
            const float m = a.aux0[dst];
            a.out_re[dst] = re * cs + im * sn;
            a.out_im[dst] = m * (im * cs - re * sn);
          }
        }
      }
    }
  }
}

__global__ void uir_magpha_fwd_kernel(const float* __restrict__ re, const float* __restrict__ im, float* __restrict__ mag,
                                      float* __restrict__ pha, int H, int Wd, int C, long n) {
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < n; idx += (long)gridDim.x * blockDim.x) {
    const long pix = idx / C;
    const int w = (int)(pix % Wd);
    const int h = (int)((pix / Wd) % H);
    float x = re[idx], y = im[idx];
    const bool hs = (h == 0) || (((H & 1) == 0) && h == H / 2);
    const bool ws = (w == 0) || (((Wd & 1) == 0) && w == Wd / 2);
    if (hs && ws) y = 0.0f;  // self-conjugate bin of a real signal: Im is exactly +0
    mag[idx] = sqrtf(x * x + y * y);
    pha[idx] = atan2f(y, x);
  }
}

__global__ void uir_magpha_bwd_kernel(const float* __restrict__ mag, const float* __restrict__ pha, const float* __restrict__ gm,
                                      const float* __restrict__ gp, float* __restrict__ gre, float* __restrict__ gim, long n) {
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < n; idx += (long)gridDim.x * blockDim.x) {
    const float m = mag[idx];
    float s, c;
    sincosf(pha[idx], &s, &c);
    float a = 0.f, b = 0.f;
    if (m > 0.f) {
      const float q = gp[idx] / m;
      a = gm[idx] * c - q * s;   // d|z|/dre = re/|z|,  dangle/dre = -im/|z|^2
      b = gm[idx] * s + q * c;   // d|z|/dim = im/|z|,  dangle/dim =  re/|z|^2
    }
    gre[idx] = a;
    gim[idx] = b;
  }
}

__global__ void uir_polar_fwd_kernel(const float* __restrict__ mag, const float* __restrict__ pha, float* __restrict__ zr,
                                     float* __restrict__ zi, long n) {
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < n; idx += (long)gridDim.x * blockDim.x) {
    float s, c;
    sincosf(pha[idx], &s, &c);
    const float m = mag[idx];
    zr[idx] = m * c;
    zi[idx] = m * s;
  }
}

__global__ void uir_polar_bwd_kernel(const float* __restrict__ mag, const float* __restrict__ pha, const float* __restrict__ gzr,
                                     const float* __restrict__ gzi, float* __restrict__ gm, float* __restrict__ gp, long n) {
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < n; idx += (long)gridDim.x * blockDim.x) {
    float s, c;
    sincosf(pha[idx], &s, &c);
    const float m = mag[idx];
    gm[idx] = gzr[idx] * c + gzi[idx] * s;
    gp[idx] = m * (gzi[idx] * c - gzr[idx] * s);
  }
}

__global__ void uir_cabs_fwd_kernel(const float* __restrict__ zr, const float* __restrict__ zi, float* __restrict__ t, long n) {
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < n; idx += (long)gridDim.x * blockDim.x) {
    const float x = zr[idx], y = zi[idx];
    t[idx] = sqrtf(x * x + y * y);
  }
}

__global__ void uir_cabs_bwd_kernel(const float* __restrict__ zr, const float* __restrict__ zi, const float* __restrict__ t,
                                    const float* __restrict__ gt, float* __restrict__ gzr, float* __restrict__ gzi, long n) {
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < n; idx += (long)gridDim.x * blockDim.x) {
    const float m = t[idx];
    const float q = m > 0.f ? gt[idx] / m : 0.f;
    gzr[idx] = q * zr[idx];
    gzi[idx] = q * zi[idx];
  }
}

inline dim3 ew_grid(long n) {
  long b = (n + 255) / 256;
  if (b > 8192) b = 8192;
  if (b < 1) b = 1;
  return dim3((unsigned)b);
}

}  // namespace

extern "C" int uir_dft_axis(const UirDftArgs* p, void* stream) {
  const UirDftArgs& a = *p;
  UIR_REQUIRE(a.in_re && a.out_re && a.cos_t && a.sin_t);
  UIR_REQUIRE(a.O > 0 && a.N > 0 && a.I > 0 && a.C > 0);
  const int ctiles = (a.C + 31) / 32;
  const int kgroups = (a.N + 32 * KT - 1) / (32 * KT);
  const long items = (long)a.O * a.I * ctiles * kgroups;
  const long blocks = (items + 3) / 4;
  UIR_REQUIRE(blocks < (1L << 31));
  hipLaunchKernelGGL(uir_dft_kernel, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream), a, ctiles, kgroups, items);
  return uir_launch_status();
}

extern "C" int uir_magpha_fwd(const float* re, const float* im, float* mag, float* pha, int B, int H, int Wd, int C, void* stream) {
  UIR_REQUIRE(re && im && mag && pha && B > 0 && H > 0 && Wd > 0 && C > 0);
  const long n = (long)B * H * Wd * C;
  hipLaunchKernelGGL(uir_magpha_fwd_kernel, ew_grid(n), dim3(256), 0, static_cast<hipStream_t>(stream), re, im, mag, pha, H, Wd, C, n);
  return uir_launch_status();
}
extern "C" int uir_magpha_bwd(const float* mag, const float* pha, const float* g_mag, const float* g_pha, float* g_re, float* g_im, long n, void* stream) {
  UIR_REQUIRE(mag && pha && g_mag && g_pha && g_re && g_im && n > 0);
  hipLaunchKernelGGL(uir_magpha_bwd_kernel, ew_grid(n), dim3(256), 0, static_cast<hipStream_t>(stream), mag, pha, g_mag, g_pha, g_re, g_im, n);
  return uir_launch_status();
}
extern "C" int uir_polar_fwd(const float* mag, const float* pha, float* zr, float* zi, long n, void* stream) {
  UIR_REQUIRE(mag && pha && zr && zi && n > 0);
  hipLaunchKernelGGL(uir_polar_fwd_kernel, ew_grid(n), dim3(256), 0, static_cast<hipStream_t>(stream), mag, pha, zr, zi, n);
  return uir_launch_status();
}
extern "C" int uir_polar_bwd(const float* mag, const float* pha, const float* g_zr, const float* g_zi, float* g_mag, float* g_pha, long n, void* stream) {
  UIR_REQUIRE(mag && pha && g_zr && g_zi && g_mag && g_pha && n > 0);
  hipLaunchKernelGGL(uir_polar_bwd_kernel, ew_grid(n), dim3(256), 0, static_cast<hipStream_t>(stream), mag, pha, g_zr, g_zi, g_mag, g_pha, n);
  return uir_launch_status();
}
extern "C" int uir_cabs_fwd(const float* zr, const float* zi, float* t, long n, void* stream) {
  UIR_REQUIRE(zr && zi && t && n > 0);
  hipLaunchKernelGGL(uir_cabs_fwd_kernel, ew_grid(n), dim3(256), 0, static_cast<hipStream_t>(stream), zr, zi, t, n);
  return uir_launch_status();
}
extern "C" int uir_cabs_bwd(const float* zr, const float* zi, const float* t, const float* g_t, float* g_zr, float* g_zi, long n, void* stream) {
  UIR_REQUIRE(zr && zi && t && g_t && g_zr && g_zi && n > 0);
  hipLaunchKernelGGL(uir_cabs_bwd_kernel, ew_grid(n), dim3(256), 0, static_cast<hipStream_t>(stream), zr, zi, t, g_t, g_zr, g_zi, n);
  return uir_launch_status();
}

extern "C" int uir_dft_axis2(const UirDft2Args* p, void* stream) {
  const UirDft2Args& a = *p;
  UIR_REQUIRE(a.in_re && a.out_re && a.ta && a.tb);
  UIR_REQUIRE(a.O > 0 && a.I > 0 && a.C > 0 && (a.N == 32 || a.N == 64 || a.N == 128));
  const int ctiles = (a.C + 31) / 32;
  const long items = (long)a.O * a.I * ctiles;
  const long blocks = (items + 3) / 4;
  UIR_REQUIRE(blocks < (1L << 31));
  hipStream_t st = static_cast<hipStream_t>(stream);
  UIR_REQUIRE(a.pro >= 0 && a.pro <= 3 && a.epi >= 0 && a.epi <= 3 && (a.pro == 0 || a.epi == 0));
  if (a.pro == UIR_DFT_PRO_POLAR) UIR_REQUIRE(a.in_im);
  if (a.pro == UIR_DFT_PRO_CABS_BWD) UIR_REQUIRE(a.aux0 && a.aux1 && a.aux2);
  if (a.pro == UIR_DFT_PRO_MAGPHA_BWD) UIR_REQUIRE(a.in_im && a.aux0 && a.aux1);
  if (a.epi != UIR_DFT_EPI_NONE) UIR_REQUIRE(a.out_im);
  if (a.epi == UIR_DFT_EPI_CABS) UIR_REQUIRE(a.aux_out);
  if (a.epi == UIR_DFT_EPI_POLAR_BWD) UIR_REQUIRE(a.aux0 && a.aux1);
  const dim3 g((unsigned)blocks), b(256);
#define UIR_D2(N2V, PV, EV) hipLaunchKernelGGL((uir_dft2_kernel<N2V, PV, EV>), g, b, 0, st, a, ctiles, items)
#define UIR_D2N(PV, EV) do { if (a.N == 32) UIR_D2(2, PV, EV); else if (a.N == 64) UIR_D2(4, PV, EV); else UIR_D2(8, PV, EV); } while (0)
  if (a.pro == UIR_DFT_PRO_POLAR) UIR_D2N(UIR_DFT_PRO_POLAR, 0);
  else if (a.pro == UIR_DFT_PRO_CABS_BWD) UIR_D2N(UIR_DFT_PRO_CABS_BWD, 0);
  else if (a.pro == UIR_DFT_PRO_MAGPHA_BWD) UIR_D2N(UIR_DFT_PRO_MAGPHA_BWD, 0);
  else if (a.epi == UIR_DFT_EPI_MAGPHA) UIR_D2N(0, UIR_DFT_EPI_MAGPHA);
  else if (a.epi == UIR_DFT_EPI_CABS) UIR_D2N(0, UIR_DFT_EPI_CABS);
  else if (a.epi == UIR_DFT_EPI_POLAR_BWD) UIR_D2N(0, UIR_DFT_EPI_POLAR_BWD);
  else UIR_D2N(0, 0);
#undef UIR_D2N
#undef UIR_D2
  return uir_launch_status();
}
