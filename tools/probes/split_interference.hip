// Probe for the interference recorded in DESIGN.md section 17: does a PURE-VALU kernel (register-only FMA chains, no LDS, constant inputs)
// lose results when the split-bf16 weight-gradient contraction (uir_gram -> uir_tiled_split_kernel<false,false>, UIR_TILED_SPLIT=1) runs
// beside it on another stream?  Links the product's C ABI.
// build + run (GPU box):  hipcc --offload-arch=gfx950 -O3 -Iinclude -o /tmp/split_interference tools/probes/split_interference.hip \
//                            -Lunderwater_image_restoration_amd -luir_hip -Wl,-rpath,$PWD/underwater_image_restoration_amd
//                         UIR_TILED_SPLIT=1 /tmp/split_interference ; /tmp/split_interference
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
#include "uir_kernels.h"

template <int NACC>
__global__ __launch_bounds__(256) void valu_chain(const float* __restrict__ in, float* __restrict__ out, int iters) {
  float acc[NACC], x[8];
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
#pragma unroll
  for (int i = 0; i < 8; ++i) x[i] = in[(t * 8 + i) & 65535];
#pragma unroll
  for (int i = 0; i < NACC; ++i) acc[i] = 0.001f * i;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_fmaf(acc[i], 0.999f, x[(i + it) & 7]);
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NACC; ++i) s += acc[i];
  out[t] = s;
}

typedef float f32x2 __attribute__((ext_vector_type(2)));
// the same chains on PACKED f32 FMAs (v_pk_fma_f32: two components per instruction, like the depthwise weight-gradient kernel)
template <int NACC>
__global__ __launch_bounds__(256) void valu_chain_pk(const float* __restrict__ in, float* __restrict__ out, int iters) {
  f32x2 acc[NACC], x[8];
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
#pragma unroll
  for (int i = 0; i < 8; ++i) x[i] = f32x2{in[(t * 16 + 2 * i) & 65535], in[(t * 16 + 2 * i + 1) & 65535]};
#pragma unroll
  for (int i = 0; i < NACC; ++i) acc[i] = f32x2{0.001f * i, 0.002f * i};
  const f32x2 k = f32x2{0.999f, 0.998f};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_elementwise_fma(acc[i], k, x[(i + it) & 7]);
  }
  f32x2 s = f32x2{0.f, 0.f};
#pragma unroll
  for (int i = 0; i < NACC; ++i) s += acc[i];
  out[t] = s.x + 3.f * s.y;
}

int main() {
  const int M = 16384, N = 2048, K = 512, rpc = 1368, nch = 12;
  float *A, *B, *slab, *in, *out, *ref;
  hipMalloc(&A, (size_t)M * N * 4); hipMalloc(&B, (size_t)M * K * 4); hipMalloc(&slab, (size_t)nch * N * K * 4);
  hipMemset(A, 0x3c, (size_t)M * N * 4); hipMemset(B, 0x3c, (size_t)M * K * 4);
  const int blocks = 128, n = blocks * 256;
  hipMalloc(&in, 65536 * 4); hipMalloc(&out, n * 4); hipMalloc(&ref, n * 4);
  std::vector<float> h(65536);
  for (int i = 0; i < 65536; ++i) h[i] = 0.25f + (i % 97) * 0.01f;
  hipMemcpy(in, h.data(), 65536 * 4, hipMemcpyHostToDevice);
  UirGramArgs g; memset(&g, 0, sizeof g);
  g.A = A; g.B = B; g.slab = slab; g.lda = N; g.ldb = K; g.a_bs = (long)M * N; g.b_bs = (long)M * K;
  g.o_sn = K; g.o_sk = 1; g.slab_stride = (long)N * K; g.sq_a_off = g.sq_b_off = g.cs_a_off = -1;
  g.M = M; g.N = N; g.K = K; g.batches = 1; g.taps = 1; g.rows_per_chunk = rpc; g.nchunks = nch;
  hipStream_t sa, sb; hipStreamCreate(&sa); hipStreamCreate(&sb);
  std::vector<float> r(n), c(n);
  for (int variant = 0; variant < 4; ++variant) {
    const int iters = 4000;
    auto victim = [&](float* dst) {
      if (variant == 0) hipLaunchKernelGGL(valu_chain<64>, dim3(blocks), dim3(256), 0, sb, in, dst, iters);
      else if (variant == 1) hipLaunchKernelGGL(valu_chain<200>, dim3(blocks), dim3(256), 0, sb, in, dst, iters / 3);
      else if (variant == 2) hipLaunchKernelGGL(valu_chain_pk<32>, dim3(blocks), dim3(256), 0, sb, in, dst, iters);
      else hipLaunchKernelGGL(valu_chain_pk<100>, dim3(blocks), dim3(256), 0, sb, in, dst, iters / 3);
    };
    victim(ref); hipDeviceSynchronize();
    hipMemcpy(r.data(), ref, n * 4, hipMemcpyDeviceToHost);
    int bad = 0, rc = 0;
    for (int it = 0; it < 100; ++it) {
      hipMemset(out, 0, n * 4);
      rc |= uir_gram(&g, sa);
      victim(out);
      hipDeviceSynchronize();
      hipMemcpy(c.data(), out, n * 4, hipMemcpyDeviceToHost);
      int diff = 0;
      for (int i = 0; i < n; ++i) diff += c[i] != r[i];
      if (diff) { ++bad; if (bad <= 3) printf("  variant %d iteration %d: %d outputs differ\n", variant, it, diff); }
    }
    const char* nm[] = {"v_fma_f32, 64 accumulators", "v_fma_f32, 200 accumulators", "v_pk_fma_f32, 32 x 2 accumulators", "v_pk_fma_f32, 100 x 2 accumulators"};
    printf("pure-VALU victim (%s per lane): %d of 100 runs differ beside uir_gram (rc %d)\n", nm[variant], bad, rc);
  }
  return 0;
}
