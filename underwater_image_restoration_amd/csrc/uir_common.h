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

static inline int uir_launch_status() {
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? UIR_OK : UIR_ELAUNCH;
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
