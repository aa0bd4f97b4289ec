// uir_attn_fwd / uir_attn_bwd: the tiny per-batch part of MDTA channel attention.
// All pixel-sized work (q.k^T Gram, attn@v, the 1x1 projections) runs in uir_gram / uir_gemm;
// what is left per batch is O(C^2 d) arithmetic on [C x C] matrices (C <= 128, d = C/heads):
// normalise + temperature + softmax, compose the attention maps with the projections
// (forward), and the chain rule back to the Gram matrix, the norms, the temperature and the
// projection weights (backward).  One workgroup (256-1024 threads) per batch sample, LDS-resident.
#include "uir_common.h"
#include "../../include/uir_kernels.h"

namespace {

constexpr int MAXCD = 2048;  // C*d: 128*16 or 32*32
constexpr float NORM_EPS = 1e-12f;  // F.normalize eps (SpectralTransformer.py:98)

__global__ __launch_bounds__(512) void uir_attn_fwd_kernel(const UirAttnFwdArgs a) {
  __shared__ float at[MAXCD];
  const int b = blockIdx.x, C = a.C, d = C / a.heads;
  const float* G = a.G + (long)b * C * C;
  const float* sq_q = a.sq_q + (long)b * C;
  const float* sq_k = a.sq_k + (long)b * C;
  for (int t = threadIdx.x; t < C; t += blockDim.x) {
    const int h = t / d;
    const float nq = fmaxf(sqrtf(sq_q[t]), NORM_EPS);
    const float T = a.temperature[h];
    float mx = -INFINITY;
    for (int j = 0; j < d; ++j) {
      const float nk = fmaxf(sqrtf(sq_k[h * d + j]), NORM_EPS);
      const float l = G[(long)t * C + h * d + j] / (nq * nk) * T;
      at[t * d + j] = l;
      mx = fmaxf(mx, l);
    }
    float s = 0.f;
    for (int j = 0; j < d; ++j) { const float e = expf(at[t * d + j] - mx); at[t * d + j] = e; s += e; }
    const float inv = 1.0f / s;
    for (int j = 0; j < d; ++j) {
      const float p = at[t * d + j] * inv;
      at[t * d + j] = p;
      a.attn[((long)b * C + t) * d + j] = p;
    }
  }
  __syncthreads();
  float* WA = a.WA + (long)b * C * C;
  float* WB = a.WB + (long)b * C * C;
  for (int e = threadIdx.x; e < C * C; e += blockDim.x) {
    const int r = e / C, col = e - r * C;
    const int h = col / d, j = col - h * d;
    float sa = 0.f, sb = 0.f;
    for (int i = 0; i < d; ++i) {
      const float p = at[(h * d + i) * d + j];
      sa = fmaf(a.P[(long)r * C + h * d + i], p, sa);
      sb = fmaf(a.Wof[(long)r * C + h * d + i], p, sb);
    }
    WA[e] = sa;
    WB[e] = sb;
  }
}

__global__ __launch_bounds__(1024) void uir_attn_bwd_kernel(const UirAttnBwdArgs a) {
  __shared__ float at[MAXCD], dA[MAXCD], S[MAXCD], dS[MAXCD];
  __shared__ float acoef[128], bcoef[128], rowdT[128], nqs[128], nks[128];
  const int b = blockIdx.x, C = a.C, d = C / a.heads;
  const float* G = a.G + (long)b * C * C;
  const float* dWA = a.dWA + (long)b * C * C;
  const float* dWB = a.dWB + (long)b * C * C;
  for (int t = threadIdx.x; t < C; t += blockDim.x) {
    nqs[t] = fmaxf(sqrtf(a.sq_q[(long)b * C + t]), NORM_EPS);
    nks[t] = fmaxf(sqrtf(a.sq_k[(long)b * C + t]), NORM_EPS);
  }
  __syncthreads();
  for (int e = threadIdx.x; e < C * d; e += blockDim.x) {
    const int t = e / d, j = e - t * d, h = t / d;
    at[e] = a.attn[(long)b * C * d + e];
    S[e] = G[(long)t * C + h * d + j] / (nqs[t] * nks[h * d + j]);
    // dA[t][j] = sum_r P[r][t] dWA[r][hd+j] + Wof[r][t] dWB[r][hd+j]
    float s = 0.f;
    for (int r = 0; r < C; ++r) {
      s = fmaf(a.P[(long)r * C + t], dWA[(long)r * C + h * d + j], s);
      s = fmaf(a.Wof[(long)r * C + t], dWB[(long)r * C + h * d + j], s);
    }
    dA[e] = s;
  }
  __syncthreads();
  for (int t = threadIdx.x; t < C; t += blockDim.x) {
    const int h = t / d;
    const float T = a.temperature[h];
    float dot = 0.f;
    for (int j = 0; j < d; ++j) dot = fmaf(dA[t * d + j], at[t * d + j], dot);
    float dt = 0.f, an = 0.f;
    for (int j = 0; j < d; ++j) {
      const float dl = at[t * d + j] * (dA[t * d + j] - dot);
      dt = fmaf(dl, S[t * d + j], dt);
      const float ds = dl * T;
      dS[t * d + j] = ds;
      an = fmaf(ds, S[t * d + j], an);
    }
    rowdT[t] = dt;
    // norm clamp inactive -> gradient flows through ||q||; clamp active -> constant denominator
    acoef[t] = (sqrtf(a.sq_q[(long)b * C + t]) > NORM_EPS) ? -an / (nqs[t] * nqs[t]) : 0.f;
  }
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    const int h = c / d, j = c - h * d;
    float bn = 0.f;
    for (int i = 0; i < d; ++i) bn = fmaf(dS[(h * d + i) * d + j], S[(h * d + i) * d + j], bn);
    bcoef[c] = (sqrtf(a.sq_k[(long)b * C + c]) > NORM_EPS) ? -bn / (nks[c] * nks[c]) : 0.f;
  }
  for (int h = threadIdx.x; h < a.heads; h += blockDim.x) {
    float s = 0.f;
    for (int i = 0; i < d; ++i) s += rowdT[h * d + i];
    a.dT[(long)b * a.heads + h] = s;
  }
  __syncthreads();
  float* M = a.Mqk + (long)b * 4 * C * C;
  const int C2 = 2 * C;
  for (int e = threadIdx.x; e < C2 * C2; e += blockDim.x) {
    const int row = e / C2, col = e - row * C2;
    float v = 0.f;
    if (row < C && col < C) {
      v = (row == col) ? acoef[row] : 0.f;
    } else if (row < C) {
      const int kc = col - C;
      if (kc / d == row / d) v = dS[row * d + (kc % d)] / (nqs[row] * nks[kc]);
    } else if (col < C) {
      const int kr = row - C;
      if (kr / d == col / d) v = dS[col * d + (kr % d)] / (nqs[col] * nks[kr]);
    } else {
      v = (row == col) ? bcoef[row - C] : 0.f;
    }
    M[e] = v;
  }
  float* dP = a.dP + (long)b * C * C;
  float* dWof = a.dWof + (long)b * C * C;
  for (int e = threadIdx.x; e < C * C; e += blockDim.x) {
    const int r = e / C, col = e - r * C;
    const int h = col / d, i = col - h * d;
    float sp = 0.f, so = 0.f;
    for (int j = 0; j < d; ++j) {
      const float p = at[(h * d + i) * d + j];
      sp = fmaf(dWA[(long)r * C + h * d + j], p, sp);
      so = fmaf(dWB[(long)r * C + h * d + j], p, so);
    }
    dP[e] = sp;
    dWof[e] = so;
  }
}

}  // namespace

extern "C" int uir_attn_fwd(const UirAttnFwdArgs* p, void* stream) {
  const UirAttnFwdArgs& a = *p;
  UIR_REQUIRE(a.G && a.sq_q && a.sq_k && a.temperature && a.P && a.Wof && a.attn && a.WA && a.WB);
  UIR_REQUIRE(a.B > 0 && a.C > 0 && a.heads > 0 && a.C % a.heads == 0 && a.C <= 128);
  UIR_REQUIRE(a.C * (a.C / a.heads) <= MAXCD);
  hipLaunchKernelGGL(uir_attn_fwd_kernel, dim3(a.B), dim3(a.C >= 64 ? 512 : 256), 0, static_cast<hipStream_t>(stream), a);
  return uir_launch_status();
}

extern "C" int uir_attn_bwd(const UirAttnBwdArgs* p, void* stream) {
  const UirAttnBwdArgs& a = *p;
  UIR_REQUIRE(a.dWA && a.dWB && a.G && a.sq_q && a.sq_k && a.temperature && a.attn && a.P && a.Wof);
  UIR_REQUIRE(a.Mqk && a.dP && a.dWof && a.dT);
  UIR_REQUIRE(a.B > 0 && a.C > 0 && a.heads > 0 && a.C % a.heads == 0 && a.C <= 128);
  UIR_REQUIRE(a.C * (a.C / a.heads) <= MAXCD);
  hipLaunchKernelGGL(uir_attn_bwd_kernel, dim3(a.B), dim3(a.C >= 64 ? 1024 : 256), 0, static_cast<hipStream_t>(stream), a);
  return uir_launch_status();
}
