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

// Both kernels are pure latency (16-128 workgroups, a few hundred flops per thread) and sit on the backward / forward
// chain 24 times per step, so they are written for the shortest dependent chain: every global operand is staged into
// LDS with coalesced loads by all threads first (the head's d columns of P / Wof / dWA / dWB as [r][i]), row and
// column reductions use 8 lanes per row with xor-shuffles instead of one thread per row, and the O(C d^2) products
// read LDS only.  Dynamic LDS: forward (d^2 + 2 C d + 64) floats, backward (4 d^2 + 4 C d + 160) floats (<= 82 KB).

__device__ __forceinline__ float sum8(float v) {
  v += __shfl_xor(v, 1);
  v += __shfl_xor(v, 2);
  v += __shfl_xor(v, 4);
  return v;
}

// grid = (batch, head): one workgroup owns one d x d attention map and the head's d columns of WA / WB
__global__ __launch_bounds__(1024) void uir_attn_fwd_kernel(const UirAttnFwdArgs a) {
  extern __shared__ float sm[];
  const int b = blockIdx.x, h = blockIdx.y, C = a.C, d = C / a.heads, hd = h * d, tid = threadIdx.x, nthr = blockDim.x;
  float* at = sm;
  float* Ps = at + d * d;
  float* Ws = Ps + C * d;
  float* nq = Ws + C * d;
  float* nk = nq + 32;
  const float* G = a.G + (long)b * C * C;
  const float T = a.temperature[h];
  for (int e = tid; e < C * d; e += nthr) {
    const int r = e / d, i = e - r * d;
    Ps[e] = a.P[(long)r * C + hd + i];
    Ws[e] = a.Wof[(long)r * C + hd + i];
  }
  if (a.slab) {                       // sum the Gram partial rows here: norms first (needed by every logit)
    if (tid < 2 * d) {
      const int isk = tid >= d, t = hd + tid - isk * d;
      const float* sp = a.slab + (isk ? a.sq_k_off : a.sq_q_off) + (long)b * C + t;
      float sv = 0.f;
#pragma unroll 4
      for (int c = 0; c < a.nch; ++c) sv += sp[(long)c * a.slab_stride];
      (isk ? a.sq_k : a.sq_q)[(long)b * C + t] = sv;
      (isk ? nk : nq)[tid - isk * d] = fmaxf(sqrtf(sv), NORM_EPS);
    }
  } else if (tid < d) {
    nq[tid] = fmaxf(sqrtf(a.sq_q[(long)b * C + hd + tid]), NORM_EPS);
    nk[tid] = fmaxf(sqrtf(a.sq_k[(long)b * C + hd + tid]), NORM_EPS);
  }
  __syncthreads();
  for (int e = tid; e < d * d; e += nthr) {
    const int i = e / d, j = e - i * d;
    const long gi = (long)(hd + i) * C + hd + j;
    float gv;
    if (a.slab) {
      const float* sp = a.slab + (long)b * C * C + gi;
      float s0 = 0.f, s1 = 0.f;
      int c = 0;
      for (; c + 1 < a.nch; c += 2) {
        s0 += sp[(long)c * a.slab_stride];
        s1 += sp[(long)(c + 1) * a.slab_stride];
      }
      if (c < a.nch) s0 += sp[(long)c * a.slab_stride];
      gv = s0 + s1;
      a.G[(long)b * C * C + gi] = gv;          // kept for uir_attn_bwd
    } else {
      gv = G[gi];
    }
    at[e] = gv / (nq[i] * nk[j]) * T;
  }
  __syncthreads();
  {
    const int row = tid >> 3, sub = tid & 7;      // 8 lanes per row; d <= 32 rows
    if (row < d) {
      float* ar = at + row * d;
      float mx = -INFINITY;
      for (int j = sub; j < d; j += 8) mx = fmaxf(mx, ar[j]);
      mx = fmaxf(mx, __shfl_xor(mx, 1));
      mx = fmaxf(mx, __shfl_xor(mx, 2));
      mx = fmaxf(mx, __shfl_xor(mx, 4));
      float sv = 0.f;
      for (int j = sub; j < d; j += 8) { const float e = expf(ar[j] - mx); ar[j] = e; sv += e; }
      const float inv = 1.0f / sum8(sv);
      float* out = a.attn + ((long)b * C + hd + row) * d;
      for (int j = sub; j < d; j += 8) { const float pv = ar[j] * inv; ar[j] = pv; out[j] = pv; }
    }
  }
  __syncthreads();
  float* WA = a.WA + (long)b * C * C;
  float* WB = a.WB + (long)b * C * C;
  for (int e = tid; e < C * d; e += nthr) {
    const int r = e / d, j = e - r * d;
    float sa = 0.f, sb = 0.f;
#pragma unroll 4
    for (int i = 0; i < d; ++i) {
      const float pv = at[i * d + j];
      sa = fmaf(Ps[r * d + i], pv, sa);
      sb = fmaf(Ws[r * d + i], pv, sb);
    }
    WA[(long)r * C + hd + j] = sa;
    WB[(long)r * C + hd + j] = sb;
  }
}

// grid = (batch, head).  Everything is separable by head: the d x d blocks of dA/dS, the head's
// 2d rows of Mqk (all 2C columns, zeros outside the head), its d columns of dP / dWof, its dT.
__global__ __launch_bounds__(1024) void uir_attn_bwd_kernel(const UirAttnBwdArgs a) {
  extern __shared__ float sm[];
  const int b = blockIdx.x, h = blockIdx.y, C = a.C, d = C / a.heads, hd = h * d, tid = threadIdx.x, nthr = blockDim.x;
  float* at = sm;
  float* dA = at + d * d;
  float* S = dA + d * d;
  float* dS = S + d * d;
  float* Ps = dS + d * d;            // this head's d columns of P / Wof / dWA / dWB as [r][i]
  float* Ws = Ps + C * d;
  float* wA = Ws + C * d;
  float* wB = wA + C * d;
  float* nqs = wB + C * d;
  float* nks = nqs + 32;
  float* acoef = nks + 32;
  float* bcoef = acoef + 32;
  float* rowdT = bcoef + 32;
  const float* G = a.G + (long)b * C * C;
  const float T = a.temperature[h];
  for (int e = tid; e < C * d; e += nthr) {
    const int r = e / d, j = e - r * d;
    const long w = (long)r * C + hd + j;
    Ps[e] = a.P[w];
    Ws[e] = a.Wof[w];
    if (a.slab) {                     // dWA / dWB arrive as uir_gram partial rows: summed here, no reduction launch
      const float* pa = a.slab + a.dWA_off + (long)b * C * C + w;
      const float* pb = a.slab + a.dWB_off + (long)b * C * C + w;
      float sa = 0.f, sb = 0.f;
#pragma unroll 4
      for (int c = 0; c < a.nch; ++c) {
        sa += pa[(long)c * a.slab_stride];
        sb += pb[(long)c * a.slab_stride];
      }
      wA[e] = sa;
      wB[e] = sb;
    } else {
      wA[e] = a.dWA[(long)b * C * C + w];
      wB[e] = a.dWB[(long)b * C * C + w];
    }
  }
  if (tid < d) {
    nqs[tid] = fmaxf(sqrtf(a.sq_q[(long)b * C + hd + tid]), NORM_EPS);
    nks[tid] = fmaxf(sqrtf(a.sq_k[(long)b * C + hd + tid]), NORM_EPS);
  }
  __syncthreads();
  for (int e = tid; e < d * d; e += nthr) {
    const int i = e / d, j = e - i * d;
    at[e] = a.attn[((long)b * C + hd + i) * d + j];
    S[e] = G[(long)(hd + i) * C + hd + j] / (nqs[i] * nks[j]);
    // dA[i][j] = sum_r P[r][hd+i] dWA[r][hd+j] + Wof[r][hd+i] dWB[r][hd+j]
    float sv = 0.f;
#pragma unroll 4
    for (int r = 0; r < C; ++r) {
      sv = fmaf(Ps[r * d + i], wA[r * d + j], sv);
      sv = fmaf(Ws[r * d + i], wB[r * d + j], sv);
    }
    dA[e] = sv;
  }
  __syncthreads();
  {
    const int row = tid >> 3, sub = tid & 7;      // softmax / temperature / normalisation backward: 8 lanes per row
    if (row < d) {
      const int o = row * d;
      float dot = 0.f;
      for (int j = sub; j < d; j += 8) dot = fmaf(dA[o + j], at[o + j], dot);
      dot = sum8(dot);
      float dt = 0.f, an = 0.f;
      for (int j = sub; j < d; j += 8) {
        const float dl = at[o + j] * (dA[o + j] - dot);
        dt = fmaf(dl, S[o + j], dt);
        const float ds = dl * T;
        dS[o + j] = ds;
        an = fmaf(ds, S[o + j], an);
      }
      dt = sum8(dt);
      an = sum8(an);
      if (sub == 0) {
        rowdT[row] = dt;
        // norm clamp inactive -> gradient flows through ||q||; clamp active -> constant denominator
        acoef[row] = (nqs[row] > NORM_EPS) ? -an / (nqs[row] * nqs[row]) : 0.f;     // nqs = max(||q||, eps)
      }
    }
  }
  __syncthreads();
  {
    const int col = tid >> 3, sub = tid & 7;
    if (col < d) {
      float bn = 0.f;
      for (int i = sub; i < d; i += 8) bn = fmaf(dS[i * d + col], S[i * d + col], bn);
      bn = sum8(bn);
      if (sub == 0) bcoef[col] = (nks[col] > NORM_EPS) ? -bn / (nks[col] * nks[col]) : 0.f;
    }
    if (tid < 32) {                       // dT = sum of the <= 32 row terms: one shuffle tree in the first half-wave
      float sv = tid < d ? rowdT[tid] : 0.f;
      sv = sum8(sv);
      sv += __shfl_xor(sv, 8);
      sv += __shfl_xor(sv, 16);
      if (tid == 0) a.dT[(long)b * a.heads + h] = sv;
    }
  }
  __syncthreads();
  // Mqk rows of this head: q-rows hd+i (row index hd+i) and k-rows (row index C+hd+j), all 2C columns
  float* M = a.Mqk + (long)b * 4 * C * C;
  const int C2 = 2 * C;
  for (int e = tid; e < 2 * d * C2; e += nthr) {
    const int lr = e / C2, col = e - lr * C2;   // lr < d: q row, else k row
    float v = 0.f;
    int row;
    if (lr < d) {
      row = hd + lr;
      if (col == row) v = acoef[lr];
      else if (col >= C + hd && col < C + hd + d) v = dS[lr * d + (col - C - hd)] / (nqs[lr] * nks[col - C - hd]);
    } else {
      const int j = lr - d;
      row = C + hd + j;
      if (col == row) v = bcoef[j];
      else if (col >= hd && col < hd + d) v = dS[(col - hd) * d + j] / (nqs[col - hd] * nks[j]);
    }
    M[(long)row * C2 + col] = v;
  }
  float* dP = a.dP + (long)b * C * C;
  float* dWof = a.dWof + (long)b * C * C;
  for (int e = tid; e < C * d; e += nthr) {
    const int r = e / d, i = e - r * d;
    float sp = 0.f, so = 0.f;
#pragma unroll 4
    for (int j = 0; j < d; ++j) {
      const float pv = at[i * d + j];
      sp = fmaf(wA[r * d + j], pv, sp);
      so = fmaf(wB[r * d + j], pv, so);
    }
    dP[(long)r * C + hd + i] = sp;
    dWof[(long)r * C + hd + i] = so;
  }
}

}  // namespace

extern "C" int uir_attn_fwd(const UirAttnFwdArgs* p, void* stream) {
  const UirAttnFwdArgs& a = *p;
  UIR_REQUIRE(a.G && a.sq_q && a.sq_k && a.temperature && a.P && a.Wof && a.attn && a.WA && a.WB);
  if (a.slab) UIR_REQUIRE(a.nch >= 1 && a.sq_q_off >= (long)a.B * a.C * a.C && a.sq_k_off >= (long)a.B * a.C * a.C &&
                          a.slab_stride >= a.sq_q_off + (long)a.B * a.C && a.slab_stride >= a.sq_k_off + (long)a.B * a.C);
  UIR_REQUIRE(a.B > 0 && a.C > 0 && a.heads > 0 && a.C % a.heads == 0 && a.C <= 128);
  UIR_REQUIRE(a.C / a.heads <= 32);
  const int d = a.C / a.heads;
  const size_t lds = (size_t)(d * d + 2 * a.C * d + 64) * sizeof(float);
  // one wave per SIMD cannot hide the LDS latency of the d- / C-long dot products: 16 waves when there is work for them
  const int nthr = a.C * d >= 1024 ? 1024 : 256;
  hipLaunchKernelGGL(uir_attn_fwd_kernel, dim3(a.B, a.heads), dim3(nthr), lds, static_cast<hipStream_t>(stream), a);
  return uir_launch_status();
}

extern "C" int uir_attn_bwd(const UirAttnBwdArgs* p, void* stream) {
  const UirAttnBwdArgs& a = *p;
  UIR_REQUIRE((a.slab || (a.dWA && a.dWB)) && a.G && a.sq_q && a.sq_k && a.temperature && a.attn && a.P && a.Wof);
  if (a.slab) UIR_REQUIRE(a.nch >= 1 && a.dWA_off >= 0 && a.dWB_off >= 0 && a.slab_stride >= a.dWA_off + (long)a.B * a.C * a.C &&
                          a.slab_stride >= a.dWB_off + (long)a.B * a.C * a.C);
  UIR_REQUIRE(a.Mqk && a.dP && a.dWof && a.dT);
  UIR_REQUIRE(a.B > 0 && a.C > 0 && a.heads > 0 && a.C % a.heads == 0 && a.C <= 128);
  UIR_REQUIRE(a.C / a.heads <= 32);
  const int d = a.C / a.heads;
  const size_t lds = (size_t)(4 * d * d + 4 * a.C * d + 160) * sizeof(float);
  const int nthr = a.C * d >= 1024 ? 1024 : 256;
  hipLaunchKernelGGL(uir_attn_bwd_kernel, dim3(a.B, a.heads), dim3(nthr), lds, static_cast<hipStream_t>(stream), a);
  return uir_launch_status();
}
