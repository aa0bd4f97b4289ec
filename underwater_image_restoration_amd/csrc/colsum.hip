// uir_colsum_partials: per-workgroup column sums of an HWC row block (bias gradients):
// slab[block][c] = sum over the block's rows of src[m][col + c].  Thread (c, slot) strides
// over rows; lanes cover consecutive channels so loads are contiguous channel runs.
#include "uir_common.h"
#include "../../include/uir_kernels.h"

namespace {
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ src, int ld, int col, long M, int C,
                                                     float* __restrict__ slab) {
  __shared__ float red[256];
  const int slots = 256 / C;
  const int c = threadIdx.x % C, slot = threadIdx.x / C;
  float s = 0.f;
  if (slot < slots)
    for (long m = (long)blockIdx.x * slots + slot; m < M; m += (long)gridDim.x * slots) s += src[m * ld + col + c];
  red[threadIdx.x] = s;
  __syncthreads();
  if ((int)threadIdx.x < C) {
    float t = 0.f;
    for (int k = 0; k < slots; ++k) t += red[k * C + threadIdx.x];
    slab[(long)blockIdx.x * C + threadIdx.x] = t;
  }
}
}  // namespace

extern "C" int uir_colsum_partials(const float* src, int ld, int col, long M, int C, float* slab, int nblocks, void* stream) {
  UIR_REQUIRE(src && slab && M > 0 && C > 0 && C <= 256 && ld >= col + C && nblocks > 0);
  hipLaunchKernelGGL(colsum_kernel, dim3(nblocks), dim3(256), 0, static_cast<hipStream_t>(stream), src, ld, col, M, C, slab);
  return uir_launch_status();
}
