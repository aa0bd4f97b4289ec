// Element-wise pieces of the FRFN feed-forward of the NewBigFRFNModel family (reference
// src/model/block.py:249-282).  The contractions of FRFN (Linear C -> 8C, Linear 4C -> C, the partial 3x3 convolution and
// the depthwise 3x3) run on uir_gemm / uir_gram / uir_dwconv*; what is left between them is
//   t        = GELU(pre)                          after linear1 (block.py:252: nn.Sequential(nn.Linear, nn.GELU()))
//   g        = GELU(u + b_dw) * x2                u = depthwise 3x3 of the x1 half WITHOUT its bias (:253 has bias=True:
//                                                 it is added here), x2 = the other half of t  (:276-279)
// and their backward passes.  Rows are pixels (the reference's (B, L, C) token layout IS the HWC row layout), float4 along
// the channels, one pass each: HBM-bound.  GELU = exact-erf form (uir_gelu_pair, |erf error| <= 1.5e-7).
#include "uir_common.h"
#include "../../include/uir_kernels.h"

namespace {

__global__ __launch_bounds__(256) void frfn_act_fwd_kernel(const float* __restrict__ pre, int ldp, float* __restrict__ t, int ldt,
                                                           long M, int cols4) {
  const long total = M * cols4;
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
    const long m = e / cols4;
    const int c = (int)(e - m * cols4) * 4;
    const float4 v = *reinterpret_cast<const float4*>(pre + m * ldp + c);
    float4 o;
    o.x = uir_gelu(v.x); o.y = uir_gelu(v.y); o.z = uir_gelu(v.z); o.w = uir_gelu(v.w);
    *reinterpret_cast<float4*>(t + m * ldt + c) = o;
  }
}

// Per-workgroup column partials of a float4-per-thread element-wise pass: the grid stride (gridDim.x * 256 float4 slots) is a
// multiple of cols4, so a thread keeps ONE float4 column for all its rows; the 256 / cols4 threads that share it are summed
// through LDS in a fixed order and the workgroup writes one slab row [4 * cols4] (summed later by uir_reduce_slab(_batch)).
__device__ __forceinline__ void frfn_col_partials(float4 acc, int cols4, float* __restrict__ slab) {
  __shared__ float4 red[256];
  red[threadIdx.x] = acc;
  __syncthreads();
  if ((int)threadIdx.x < cols4) {
    float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int k = threadIdx.x; k < 256; k += cols4) {
      const float4 v = red[k];
      t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w;
    }
    *reinterpret_cast<float4*>(slab + (long)blockIdx.x * (4 * cols4) + 4 * threadIdx.x) = t;
  }
}

__global__ __launch_bounds__(256) void frfn_act_bwd_kernel(const float* __restrict__ pre, int ldp, const float* __restrict__ dt,
                                                           int ldt, float* __restrict__ dpre, int ldo, long M, int cols4,
                                                           float* __restrict__ db_slab) {
  const long total = M * cols4;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
    const long m = e / cols4;
    const int c = (int)(e - m * cols4) * 4;
    const float4 v = *reinterpret_cast<const float4*>(pre + m * ldp + c);
    const float4 d = *reinterpret_cast<const float4*>(dt + m * ldt + c);
    float4 o;
    o.x = d.x * uir_gelu_grad(v.x); o.y = d.y * uir_gelu_grad(v.y); o.z = d.z * uir_gelu_grad(v.z); o.w = d.w * uir_gelu_grad(v.w);
    *reinterpret_cast<float4*>(dpre + m * ldo + c) = o;
    acc.x += o.x; acc.y += o.y; acc.z += o.z; acc.w += o.w;
  }
  if (db_slab) frfn_col_partials(acc, cols4, db_slab);
}

__global__ __launch_bounds__(256) void frfn_gate_fwd_kernel(const UirFrfnGateArgs a) {
  const int cols4 = a.hid / 4;
  const long total = a.M * cols4;
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
    const long m = e / cols4;
    const int c = (int)(e - m * cols4) * 4;
    const float4 u = *reinterpret_cast<const float4*>(a.U + m * a.ldu + c);
    const float4 b = *reinterpret_cast<const float4*>(a.bias + c);
    float4 v = make_float4(1.f, 1.f, 1.f, 1.f);        // V == NULL: plain GELU(u + bias) (LeFF, block.py:228,240)
    if (a.V) v = *reinterpret_cast<const float4*>(a.V + m * a.ldv + c);
    if (a.v_act) v = make_float4(uir_gelu(v.x), uir_gelu(v.y), uir_gelu(v.z), uir_gelu(v.w));
    float4 o;
    o.x = uir_gelu(u.x + b.x) * v.x; o.y = uir_gelu(u.y + b.y) * v.y; o.z = uir_gelu(u.z + b.z) * v.z; o.w = uir_gelu(u.w + b.w) * v.w;
    *reinterpret_cast<float4*>(a.G + m * a.ldg + c) = o;
  }
}

__global__ __launch_bounds__(256) void frfn_gate_bwd_kernel(const UirFrfnGateArgs a) {
  const int cols4 = a.hid / 4;
  const long total = a.M * cols4;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f), accv = acc;
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
    const long m = e / cols4;
    const int c = (int)(e - m * cols4) * 4;
    const float4 u = *reinterpret_cast<const float4*>(a.U + m * a.ldu + c);
    const float4 b = *reinterpret_cast<const float4*>(a.bias + c);
    const float4 d = *reinterpret_cast<const float4*>(a.dG + m * a.ldg + c);
    float4 v = make_float4(1.f, 1.f, 1.f, 1.f), va = v;       // va: gelu'(V) when V is a pre-activation
    if (a.V) v = *reinterpret_cast<const float4*>(a.V + m * a.ldv + c);
    if (a.v_act) {
      const float4 p = v;
      uir_gelu_pair(p.x, v.x, va.x); uir_gelu_pair(p.y, v.y, va.y); uir_gelu_pair(p.z, v.z, va.z); uir_gelu_pair(p.w, v.w, va.w);
    }
    float4 du, dv;
    float g, dg;
    uir_gelu_pair(u.x + b.x, g, dg); du.x = d.x * v.x * dg; dv.x = d.x * g;
    uir_gelu_pair(u.y + b.y, g, dg); du.y = d.y * v.y * dg; dv.y = d.y * g;
    uir_gelu_pair(u.z + b.z, g, dg); du.z = d.z * v.z * dg; dv.z = d.z * g;
    uir_gelu_pair(u.w + b.w, g, dg); du.w = d.w * v.w * dg; dv.w = d.w * g;
    *reinterpret_cast<float4*>(a.dU + m * a.lddu + c) = du;
    dv = make_float4(dv.x * va.x, dv.y * va.y, dv.z * va.z, dv.w * va.w);
    if (a.dV) *reinterpret_cast<float4*>(a.dV + m * a.lddv + c) = dv;
    acc.x += du.x; acc.y += du.y; acc.z += du.z; acc.w += du.w;
    accv.x += dv.x; accv.y += dv.y; accv.z += dv.z; accv.w += dv.w;
  }
  if (a.db_slab) frfn_col_partials(acc, cols4, a.db_slab);
  if (a.dbv_slab) {
    __syncthreads();                                           // the helper's LDS buffer is read by the first call's last phase
    frfn_col_partials(accv, cols4, a.dbv_slab);
  }
}

// dz = dy * (y > 0 ? 1 : slope): backward of an in-place LeakyReLU given its OUTPUT (InputProjection, block.py:50)
__global__ __launch_bounds__(256) void leaky_bwd_kernel(const float* __restrict__ y, int ldy, const float* __restrict__ dy, int lddy,
                                                        float* __restrict__ dz, int lddz, long M, int cols4, float slope) {
  const long total = M * cols4;
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
    const long m = e / cols4;
    const int c = (int)(e - m * cols4) * 4;
    const float4 v = *reinterpret_cast<const float4*>(y + m * ldy + c);
    const float4 d = *reinterpret_cast<const float4*>(dy + m * lddy + c);
    *reinterpret_cast<float4*>(dz + m * lddz + c) =
        make_float4(v.x > 0.f ? d.x : d.x * slope, v.y > 0.f ? d.y : d.y * slope, v.z > 0.f ? d.z : d.z * slope, v.w > 0.f ? d.w : d.w * slope);
  }
}

inline unsigned grid_for(long total, int cap = 8192) {
  long g = (total + 255) / 256;
  return (unsigned)(g > cap ? cap : (g < 1 ? 1 : g));
}

}  // namespace

extern "C" int uir_frfn_act_fwd(const float* pre, int ldp, float* t, int ldt, long M, int cols, void* stream) {
  UIR_REQUIRE(pre && t && M > 0 && cols > 0 && (cols & 3) == 0 && ldp >= cols && ldt >= cols && (ldp & 3) == 0 && (ldt & 3) == 0);
  UIR_REQUIRE(uir_aligned16(pre) && uir_aligned16(t));
  hipLaunchKernelGGL(frfn_act_fwd_kernel, dim3(grid_for(M * (cols / 4))), dim3(256), 0, static_cast<hipStream_t>(stream), pre, ldp, t, ldt,
                     M, cols / 4);
  return uir_launch_status();
}

extern "C" int uir_frfn_act_bwd(const float* pre, int ldp, const float* dt, int ldt, float* dpre, int ldo, long M, int cols,
                                float* db_slab, int nblocks, void* stream) {
  UIR_REQUIRE(pre && dt && dpre && M > 0 && cols > 0 && (cols & 3) == 0 && ldp >= cols && ldt >= cols && ldo >= cols);
  UIR_REQUIRE((ldp & 3) == 0 && (ldt & 3) == 0 && (ldo & 3) == 0 && uir_aligned16(pre) && uir_aligned16(dt) && uir_aligned16(dpre));
  // with column partials: a thread must keep one float4 column (256 % (cols/4) == 0) and the grid is the caller's slab height
  UIR_REQUIRE(!db_slab || (nblocks > 0 && cols <= 1024 && 256 % (cols / 4) == 0 && uir_aligned16(db_slab)));
  const unsigned g = db_slab ? (unsigned)nblocks : grid_for(M * (cols / 4));
  hipLaunchKernelGGL(frfn_act_bwd_kernel, dim3(g), dim3(256), 0, static_cast<hipStream_t>(stream), pre, ldp, dt, ldt, dpre, ldo, M,
                     cols / 4, db_slab);
  return uir_launch_status();
}

static int frfn_gate_check(const UirFrfnGateArgs& a, bool bwd) {
  UIR_REQUIRE(a.U && a.bias && a.M > 0 && a.hid > 0 && (a.hid & 3) == 0);
  UIR_REQUIRE(a.ldu >= a.hid && a.ldg >= a.hid && ((a.ldu | a.ldg) & 3) == 0 && uir_aligned16(a.U) && uir_aligned16(a.bias));
  UIR_REQUIRE(!a.V || (a.ldv >= a.hid && (a.ldv & 3) == 0 && uir_aligned16(a.V)));
  if (!bwd) UIR_REQUIRE(a.G && uir_aligned16(a.G));
  else {
    UIR_REQUIRE(a.dG && a.dU && a.lddu >= a.hid && (a.lddu & 3) == 0 && uir_aligned16(a.dG) && uir_aligned16(a.dU));
    UIR_REQUIRE((a.dV != nullptr) == (a.V != nullptr) && (!a.dV || (a.lddv >= a.hid && (a.lddv & 3) == 0 && uir_aligned16(a.dV))));
    UIR_REQUIRE(!a.db_slab || (a.db_blocks > 0 && a.hid <= 1024 && 256 % (a.hid / 4) == 0 && uir_aligned16(a.db_slab)));
    UIR_REQUIRE(!a.dbv_slab || (a.db_slab && a.dV && uir_aligned16(a.dbv_slab)));
  }
  UIR_REQUIRE((a.v_act == 0 || a.v_act == 1) && (!a.v_act || a.V));
  return UIR_OK;
}

extern "C" int uir_frfn_gate_fwd(const UirFrfnGateArgs* p, void* stream) {
  const UirFrfnGateArgs& a = *p;
  if (int rc = frfn_gate_check(a, false)) return rc;
  hipLaunchKernelGGL(frfn_gate_fwd_kernel, dim3(grid_for(a.M * (a.hid / 4))), dim3(256), 0, static_cast<hipStream_t>(stream), a);
  return uir_launch_status();
}

extern "C" int uir_frfn_gate_bwd(const UirFrfnGateArgs* p, void* stream) {
  const UirFrfnGateArgs& a = *p;
  if (int rc = frfn_gate_check(a, true)) return rc;
  const unsigned g = a.db_slab ? (unsigned)a.db_blocks : grid_for(a.M * (a.hid / 4));
  hipLaunchKernelGGL(frfn_gate_bwd_kernel, dim3(g), dim3(256), 0, static_cast<hipStream_t>(stream), a);
  return uir_launch_status();
}

extern "C" int uir_leaky_bwd(const float* y, int ldy, const float* dy, int lddy, float* dz, int lddz, long M, int cols, float slope,
                             void* stream) {
  UIR_REQUIRE(y && dy && dz && M > 0 && cols > 0 && (cols & 3) == 0 && ldy >= cols && lddy >= cols && lddz >= cols);
  UIR_REQUIRE(((ldy | lddy | lddz) & 3) == 0 && uir_aligned16(y) && uir_aligned16(dy) && uir_aligned16(dz));
  hipLaunchKernelGGL(leaky_bwd_kernel, dim3(grid_for(M * (cols / 4))), dim3(256), 0, static_cast<hipStream_t>(stream), y, ldy, dy, lddy, dz,
                     lddz, M, cols / 4, slope);
  return uir_launch_status();
}
