// uir_attn_fwd / uir_attn_bwd: the tiny per-batch part of MDTA channel attention.
// All pixel-sized work (q.k^T Gram, attn@v, the 1x1 projections) runs in uir_gram / uir_gemm;
// what is left per batch is O(C^2 d) arithmetic on [C x C] matrices (C <= 128, d = C/heads):
// normalise + temperature + softmax, compose the attention maps with the projections
// (forward), and the chain rule back to the Gram matrix, the norms, the temperature and the
// projection weights (backward).  One workgroup per (batch sample, head), LDS-resident (d <= 32).
#include "uir_common.h"
#include "../../include/uir_kernels.h"

namespace {

constexpr float NORM_EPS = 1e-12f;  // F.normalize eps (SpectralTransformer.py:98)

// grid = (batch, head): one workgroup owns one d x d attention map and the head's d columns of WA / WB
__global__ __launch_bounds__(256) void uir_attn_fwd_kernel(const UirAttnFwdArgs a) {
  __shared__ float at[32 * 32];
  const int b = blockIdx.x, h = blockIdx.y, C = a.C, d = C / a.heads;
  const float* G = a.G + (long)b * C * C;
  const float* sq_q = a.sq_q + (long)b * C;
  const float* sq_k = a.sq_k + (long)b * C;
  const float T = a.temperature[h];
  for (int i = threadIdx.x; i < d; i += blockDim.x) {
    const int t = h * d + i;
    const float nq = fmaxf(sqrtf(sq_q[t]), NORM_EPS);
    float mx = -INFINITY;
    for (int j = 0; j < d; ++j) {
      const float nk = fmaxf(sqrtf(sq_k[h * d + j]), NORM_EPS);
      const float l = G[(long)t * C + h * d + j] / (nq * nk) * T;
      at[i * d + j] = l;
      mx = fmaxf(mx, l);
    }
    float s = 0.f;
    for (int j = 0; j < d; ++j) { const float e = expf(at[i * d + j] - mx); at[i * d + j] = e; s += e; }
    const float inv = 1.0f / s;
    for (int j = 0; j < d; ++j) {
      const float p = at[i * d + j] * inv;
      at[i * d + j] = p;
      a.attn[((long)b * C + t) * d + j] = p;
    }
  }
  __syncthreads();
  float* WA = a.WA + (long)b * C * C;
  float* WB = a.WB + (long)b * C * C;
  for (int e = threadIdx.x; e < C * d; e += blockDim.x) {
    const int r = e / d, j = e - r * d;
    float sa = 0.f, sb = 0.f;
    for (int i = 0; i < d; ++i) {
      const float p = at[i * d + j];
      sa = fmaf(a.P[(long)r * C + h * d + i], p, sa);
      sb = fmaf(a.Wof[(long)r * C + h * d + i], p, sb);
    }
    WA[(long)r * C + h * d + j] = sa;
    WB[(long)r * C + h * d + j] = sb;
  }
}

// grid = (batch, head).  Everything is separable by head: the d x d blocks of dA/dS, the head's
// 2d rows of Mqk (all 2C columns, zeros outside the head), its d columns of dP / dWof, its dT.
__global__ __launch_bounds__(256) void uir_attn_bwd_kernel(const UirAttnBwdArgs a) {
  __shared__ float at[1024], dA[1024], S[1024], dS[1024];
  __shared__ float acoef[32], bcoef[32], rowdT[32], nqs[32], nks[32];
  __shared__ float wA[128 * 32], wB[128 * 32];   // this head's d columns of dWA / dWB, [r][j]: read twice below
  const int b = blockIdx.x, h = blockIdx.y, C = a.C, d = C / a.heads;
  const float* G = a.G + (long)b * C * C;
  const float T = a.temperature[h];
  for (int e = threadIdx.x; e < C * d; e += blockDim.x) {
    const int r = e / d, j = e - r * d;
    const long idx = (long)b * C * C + (long)r * C + h * d + j;
    wA[e] = a.dWA[idx];
    wB[e] = a.dWB[idx];
  }
  for (int i = threadIdx.x; i < d; i += blockDim.x) {
    nqs[i] = fmaxf(sqrtf(a.sq_q[(long)b * C + h * d + i]), NORM_EPS);
    nks[i] = fmaxf(sqrtf(a.sq_k[(long)b * C + h * d + i]), NORM_EPS);
  }
  __syncthreads();
  for (int e = threadIdx.x; e < d * d; e += blockDim.x) {
    const int i = e / d, j = e - i * d, t = h * d + i;
    at[e] = a.attn[((long)b * C + t) * d + j];
    S[e] = G[(long)t * C + h * d + j] / (nqs[i] * nks[j]);
    // dA[i][j] = sum_r P[r][t] dWA[r][hd+j] + Wof[r][t] dWB[r][hd+j]
    float s = 0.f;
    for (int r = 0; r < C; ++r) {
      s = fmaf(a.P[(long)r * C + t], wA[r * d + j], s);
      s = fmaf(a.Wof[(long)r * C + t], wB[r * d + j], s);
    }
    dA[e] = s;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < d; i += blockDim.x) {
    float dot = 0.f;
    for (int j = 0; j < d; ++j) dot = fmaf(dA[i * d + j], at[i * d + j], dot);
    float dt = 0.f, an = 0.f;
    for (int j = 0; j < d; ++j) {
      const float dl = at[i * d + j] * (dA[i * d + j] - dot);
      dt = fmaf(dl, S[i * d + j], dt);
      const float ds = dl * T;
      dS[i * d + j] = ds;
      an = fmaf(ds, S[i * d + j], an);
    }
    rowdT[i] = dt;
    // norm clamp inactive -> gradient flows through ||q||; clamp active -> constant denominator
    acoef[i] = (sqrtf(a.sq_q[(long)b * C + h * d + i]) > NORM_EPS) ? -an / (nqs[i] * nqs[i]) : 0.f;
  }
  __syncthreads();
  for (int j = threadIdx.x; j < d; j += blockDim.x) {
    float bn = 0.f;
    for (int i = 0; i < d; ++i) bn = fmaf(dS[i * d + j], S[i * d + j], bn);
    bcoef[j] = (sqrtf(a.sq_k[(long)b * C + h * d + j]) > NORM_EPS) ? -bn / (nks[j] * nks[j]) : 0.f;
  }
  if (threadIdx.x == 0) {
    float s = 0.f;
    for (int i = 0; i < d; ++i) s += rowdT[i];
    a.dT[(long)b * a.heads + h] = s;
  }
  __syncthreads();
  // Mqk rows of this head: q-rows hd+i (row index hd+i) and k-rows (row index C+hd+j), all 2C columns
  float* M = a.Mqk + (long)b * 4 * C * C;
  const int C2 = 2 * C;
  for (int e = threadIdx.x; e < 2 * d * C2; e += blockDim.x) {
    const int lr = e / C2, col = e - lr * C2;   // lr < d: q row, else k row
    float v = 0.f;
    int row;
    if (lr < d) {
      row = h * d + lr;
      if (col == row) v = acoef[lr];
      else if (col >= C + h * d && col < C + h * d + d) v = dS[lr * d + (col - C - h * d)] / (nqs[lr] * nks[col - C - h * d]);
    } else {
      const int j = lr - d;
      row = C + h * d + j;
      if (col == row) v = bcoef[j];
      else if (col >= h * d && col < h * d + d) v = dS[(col - h * d) * d + j] / (nqs[col - h * d] * nks[j]);
    }
    M[(long)row * C2 + col] = v;
  }
  float* dP = a.dP + (long)b * C * C;
  float* dWof = a.dWof + (long)b * C * C;
  for (int e = threadIdx.x; e < C * d; e += blockDim.x) {
    const int r = e / d, i = e - r * d;
    float sp = 0.f, so = 0.f;
    for (int j = 0; j < d; ++j) {
      const float p = at[i * d + j];
      sp = fmaf(wA[r * d + j], p, sp);
      so = fmaf(wB[r * d + j], p, so);
    }
    dP[(long)r * C + h * d + i] = sp;
    dWof[(long)r * C + h * d + i] = so;
  }
}

}  // namespace

extern "C" int uir_attn_fwd(const UirAttnFwdArgs* p, void* stream) {
  const UirAttnFwdArgs& a = *p;
  UIR_REQUIRE(a.G && a.sq_q && a.sq_k && a.temperature && a.P && a.Wof && a.attn && a.WA && a.WB);
  UIR_REQUIRE(a.B > 0 && a.C > 0 && a.heads > 0 && a.C % a.heads == 0 && a.C <= 128);
  UIR_REQUIRE(a.C / a.heads <= 32);
  hipLaunchKernelGGL(uir_attn_fwd_kernel, dim3(a.B, a.heads), dim3(256), 0, static_cast<hipStream_t>(stream), a);
  return uir_launch_status();
}

extern "C" int uir_attn_bwd(const UirAttnBwdArgs* p, void* stream) {
  const UirAttnBwdArgs& a = *p;
  UIR_REQUIRE(a.dWA && a.dWB && a.G && a.sq_q && a.sq_k && a.temperature && a.attn && a.P && a.Wof);
  UIR_REQUIRE(a.Mqk && a.dP && a.dWof && a.dT);
  UIR_REQUIRE(a.B > 0 && a.C > 0 && a.heads > 0 && a.C % a.heads == 0 && a.C <= 128);
  UIR_REQUIRE(a.C / a.heads <= 32);
  hipLaunchKernelGGL(uir_attn_bwd_kernel, dim3(a.B, a.heads), dim3(256), 0, static_cast<hipStream_t>(stream), a);
  return uir_launch_status();
}
