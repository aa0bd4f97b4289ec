// Shared device/host helpers for the gfx950 (MI355X, CDNA4) kernels.
// wave = 64 lanes; f32 MFMA (v_mfma_f32_32x32x2_f32) for every channel contraction.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

#define UIR_OK 0
#define UIR_EINVAL 1
#define UIR_ELAUNCH 2

#define UIR_REQUIRE(cond) do { if (!(cond)) return UIR_EINVAL; } while (0)

// Status of the launch just made.  hipGetLastError is sticky since ROCm 7 (it reports the last ERROR of any HIP call on this thread, e.g. one the
// host framework provoked and handled before this entry point ran - seen as a spurious UIR_ELAUNCH on the first launch of a plan);
// hipExtGetLastError is the status of the most recent HIP call, i.e. of this entry point's own launch.
inline thread_local hipError_t uir_last_hip_error = hipSuccess;   // read by uir_last_launch_error()
static inline int uir_launch_status() {
  hipError_t e = hipExtGetLastError();
  if (e == hipSuccess) return UIR_OK;
  (void)hipGetLastError();             // our own failure: consumed here, reported through the return code
  uir_last_hip_error = e;
  return UIR_ELAUNCH;
}

static inline bool uir_aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// padded column -> real index (or -1 for an exact-zero pad column).
// Columns come in groups of grp_pad, of which the first grp_real are real.
__device__ __forceinline__ int uir_colmap(int c, int grp_real, int grp_pad) {
  if (grp_pad <= 0) return c;
  int g = c / grp_pad;
  int r = c - g * grp_pad;
  return r < grp_real ? g * grp_real + r : -1;
}

// erf-GELU (torch F.gelu default, SpectralTransformer.py:129) and its derivative from ONE
// exponential: Phi(x) via Abramowitz-Stegun 7.1.26 (|erf error| <= 1.5e-7, i.e. fp32 rounding
// level; the tails are evaluated without cancellation), phi(x) = exp(-x^2/2)/sqrt(2 pi) reuses
// the same exp.  ~15 VALU ops instead of ~60 for erff() + expf(): the gate kernels were VALU-bound.
__device__ __forceinline__ void uir_gelu_pair(float x, float& g, float& dg) {
  const float ax = fabsf(x);
  const float z = ax * 0.70710678118654752440f;
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.0f));   // v_rcp_f32 (1 ulp); __frcp_rn is a 10-instruction IEEE division
  const float e = __expf(-z * z);
  float poly = fmaf(t, 1.061405429f, -1.453152027f);
  poly = fmaf(t, poly, 1.421413741f);
  poly = fmaf(t, poly, -0.284496736f);
  poly = fmaf(t, poly, 0.254829592f);
  const float tail = 0.5f * poly * t * e;               // 1 - Phi(|x|)
  const float cdf = x >= 0.f ? 1.0f - tail : tail;
  g = x * cdf;
  dg = fmaf(x, 0.39894228040143267794f * e, cdf);
}
__device__ __forceinline__ float uir_gelu(float x) {
  float g, dg;
  uir_gelu_pair(x, g, dg);
  return g;
}
__device__ __forceinline__ float uir_gelu_grad(float x) {
  float g, dg;
  uir_gelu_pair(x, g, dg);
  return dg;
}

// row index held by accumulator register `reg` of a 32x32 f32 MFMA tile for lane half h2
__device__ __forceinline__ int uir_mfma_row(int reg, int h2) { return (reg & 3) + 8 * (reg >> 2) + 4 * h2; }

__device__ __forceinline__ f32x16 uir_mfma(float a, float b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

// ---- split-bf16 contraction in registers (the construction of csrc/tiled.hip, DESIGN.md section 17, without the LDS planes) ----
// An f32 value is the exact sum of three bf16 pieces (hi = its top 16 bits, mid = the top 16 bits of x - hi, lo = x - hi - mid, both
// subtractions exact), a bf16 x bf16 product is exact in f32, and the six products (h,h) (h,m) (m,h) (h,l) (l,h) (m,m) reproduce
// a b to 3 * 2^-24 |a b| - the rounding level of an f32 multiply - on v_mfma_f32_32x32x16_bf16, which runs at 16x the rate of
// v_mfma_f32_32x32x2_f32 and, unlike it, beside the VALU.  A lane's 8 operand values of one 16-deep k step are the SAME 8 values it
// feeds to eight consecutive f32 MFMAs (k = 8 j + 4 h2 + e for two consecutive j): any bijection lane slot <-> k is a valid
// contraction order as long as both operands use it, so the f32 kernels' operand registers are packed as they stand.
typedef __bf16 uir_bf16x8 __attribute__((ext_vector_type(8)));
struct UirBf3 { uir_bf16x8 h, m, l; };

__device__ __forceinline__ uint32_t uir_pack_hi16(uint32_t lo_elem, uint32_t hi_elem) {   // {lo_elem >> 16, hi_elem >> 16} as one dword
  return __builtin_amdgcn_perm(hi_elem, lo_elem, 0x07060302u);
}
__device__ __forceinline__ UirBf3 uir_split8(const float (&x)[8]) {
  uint32_t hh[8], mm[8], ll[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    hh[e] = __float_as_uint(x[e]) & 0xffff0000u;
    const float r1 = x[e] - __uint_as_float(hh[e]);
    mm[e] = __float_as_uint(r1) & 0xffff0000u;
    ll[e] = __float_as_uint(r1 - __uint_as_float(mm[e]));       // at most 8 significant bits left: its top 16 bits hold it exactly
  }
  typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
  UirBf3 o;
  o.h = __builtin_bit_cast(uir_bf16x8, (u32x4){uir_pack_hi16(hh[0], hh[1]), uir_pack_hi16(hh[2], hh[3]), uir_pack_hi16(hh[4], hh[5]), uir_pack_hi16(hh[6], hh[7])});
  o.m = __builtin_bit_cast(uir_bf16x8, (u32x4){uir_pack_hi16(mm[0], mm[1]), uir_pack_hi16(mm[2], mm[3]), uir_pack_hi16(mm[4], mm[5]), uir_pack_hi16(mm[6], mm[7])});
  o.l = __builtin_bit_cast(uir_bf16x8, (u32x4){uir_pack_hi16(ll[0], ll[1]), uir_pack_hi16(ll[2], ll[3]), uir_pack_hi16(ll[4], ll[5]), uir_pack_hi16(ll[6], ll[7])});
  return o;
}
__device__ __forceinline__ UirBf3 uir_split8(const float4& a, const float4& b) {
  const float x[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
  return uir_split8(x);
}
// four values -> three pairs of bf16 (one dword pair per plane), for staging threads that split a weight chunk ONCE for all the waves of a
// workgroup (each wave re-splitting the chunk it reads was 45 % of a chunk's time in uir_gdfn_fwd<128>)
struct UirBf3x4 { uint2 h, m, l; };
__device__ __forceinline__ UirBf3x4 uir_split4(const float4& v) {
  const float x[4] = {v.x, v.y, v.z, v.w};
  uint32_t hh[4], mm[4], ll[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    hh[e] = __float_as_uint(x[e]) & 0xffff0000u;
    const float r1 = x[e] - __uint_as_float(hh[e]);
    mm[e] = __float_as_uint(r1) & 0xffff0000u;
    ll[e] = __float_as_uint(r1 - __uint_as_float(mm[e]));
  }
  UirBf3x4 o;
  o.h = make_uint2(uir_pack_hi16(hh[0], hh[1]), uir_pack_hi16(hh[2], hh[3]));
  o.m = make_uint2(uir_pack_hi16(mm[0], mm[1]), uir_pack_hi16(mm[2], mm[3]));
  o.l = make_uint2(uir_pack_hi16(ll[0], ll[1]), uir_pack_hi16(ll[2], ll[3]));
  return o;
}
// c += A . B over one 16-deep k step, f32-accurate; small terms first
__device__ __forceinline__ f32x16 uir_mfma_split(const UirBf3& a, const UirBf3& b, f32x16 c) {
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.m, b.m, c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.l, b.h, c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.h, b.l, c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.m, b.h, c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.h, b.m, c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.h, b.h, c, 0, 0, 0);
  return c;
}
// UIR_FUSED_SPLIT (knobs.py), read once: 1 (default) = the tile-fused kernels contract on split-bf16, 0 = on the f32 MFMA
int uir_fused_split();

__device__ __forceinline__ float uir_wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// Internal (not part of the C ABI): the LDS-tiled wide contraction of csrc/tiled.hip, reached through uir_gemm / uir_gram.
//   D[split][i * sd_i + j * sd_j] = epilogue( sum_{k in split} A[i * sa_i + k * sa_k] * B[j * sb_j + k * sb_k] )
// Each operand has sa_k == 1 (k-contiguous) or sa_i == 1 (index-contiguous); 16-byte aligned, strides multiples of 4.
struct UirTiledArgs {
  const float* A;
  const float* B;
  float* D;
  long sa_i, sa_k, sb_j, sb_k, sd_i, sd_j;
  long d_split;            // floats between the partial outputs of consecutive k splits
  int MM, NN, KK;
  int k_per_split;         // multiple of 4; ceil(KK / k_per_split) splits
  const float* bias;       // [NN] or null
  const float* R;          // residual [MM][ldr] or null
  float* pre_out;          // pre-activation copy [MM][ldpre] or null
  long ldr, ldpre;
  int act;                 // UIR_ACT_NONE / UIR_ACT_GELU
  int allow_split;         // caller's mark for the split-bf16 variant (wide layers only), see uir_tiled_gemm
  int vec_epi;             // set by uir_tiled_gemm: D / R / pre_out rows are 16-byte addressable (row-major float4 epilogue of the 16-bit kernel)
  const float* out_scale;  // UirGemmArgs.out_scale / scale_rows (plain tiled kernel only), or null
  int scale_rows;
  int H, Wd, flip;         // uir_tiled_conv3x3: image size of the pixel rows, weight taps mirrored (input gradient)
  long sb_t;               // uir_tiled_conv3x3: tap stride of the weight
};
int uir_tiled_gemm(const UirTiledArgs& t, hipStream_t st);
// UIR_MIXED (knobs.py), read once: 0 off, 1 bf16, 2 fp16, -1 an unknown value (uir_tiled_gemm refuses it)
int uir_mixed_mode();
// 3 x 3 convolution (uir_gemm's taps = 9 contract) with K >= 128 input channels: A k-contiguous pixel rows, any weight strides
int uir_tiled_conv3x3(const UirTiledArgs& t, hipStream_t st);
