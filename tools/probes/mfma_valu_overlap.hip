// Probe: can f32 MFMA (v_mfma_f32_32x32x2_f32) and VALU work overlap on one SIMD of gfx950
//  (a) across two waves of one SIMD (one wave MFMA-only, the other VALU-only)?
//  (b) inside one wave (independent VALU issued in the shadow of an MFMA)?
// build: hipcc --offload-arch=gfx950 -O3 -o mfma_valu_overlap mfma_valu_overlap.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(256, 2) void probe(float* out, int iters, int mode, int* simd_cnt) {
  // role = arrival order on the physical SIMD (HW_ID: simd [5:4], cu [11:8], sh [12], se [14:13]; XCC_ID [3:0])
  const unsigned hw = __builtin_amdgcn_s_getreg(63492), xcc = __builtin_amdgcn_s_getreg(6164);
  const unsigned simd = ((hw >> 4) & 3) | (((hw >> 8) & 0x7f) << 2) | ((xcc & 15) << 9);
  int slot = 0;
  if ((threadIdx.x & 63) == 0) slot = atomicAdd(&simd_cnt[simd], 1);
  slot = __shfl(slot, 0) & 1;
  f32x16 acc;
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  float v[16];
  for (int i = 0; i < 16; ++i) v[i] = threadIdx.x * 0.001f + i;
  const float a = threadIdx.x * 0.5f, b = 1.0f + threadIdx.x;
  const bool do_mfma = mode == 0 || (mode == 2 && slot == 0) || mode == 3;
  const bool do_valu = mode == 1 || (mode == 2 && slot == 1) || mode == 3;
  for (int it = 0; it < iters; ++it) {
    if (do_mfma) {
#pragma unroll
      for (int k = 0; k < 4; ++k) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    }
    if (do_valu) {
      // 4 MFMAs = 256 cycles of matrix pipe; 64 VALU ops = 256 cycles of VALU issue
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int i = 0; i < 16; ++i) v[i] = __builtin_fmaf(v[i], 1.0001f, 0.5f);
    }
  }
  float s = 0.f;
  for (int i = 0; i < 16; ++i) s += acc[i] + v[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main() {
  float* out;
  const int blocks = 512, iters = 20000;
  hipMalloc(&out, blocks * 256 * sizeof(float));
  int* cnt;
  hipMalloc(&cnt, 8192 * sizeof(int));
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const char* names[] = {"all waves MFMA only", "all waves VALU only", "even slot MFMA / odd slot VALU", "every wave: MFMA + VALU interleaved"};
  for (int mode = 0; mode < 4; ++mode) {
    hipMemset(cnt, 0, 8192 * sizeof(int));
    hipLaunchKernelGGL(probe, dim3(blocks), dim3(256), 0, 0, out, 100, mode, cnt);
    hipMemset(cnt, 0, 8192 * sizeof(int));
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(probe, dim3(blocks), dim3(256), 0, 0, out, iters, mode, cnt);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    int h[8192], mx = 0, used = 0, odd = 0;
    hipMemcpy(h, cnt, sizeof(h), hipMemcpyDeviceToHost);
    for (int i = 0; i < 8192; ++i) { if (h[i]) ++used; if (h[i] > mx) mx = h[i]; if (h[i] & 1) ++odd; }
    printf("mode %d (%s): %.3f ms   [SIMDs used %d, max waves/SIMD %d, SIMDs with odd count %d]\n", mode, names[mode], ms, used, mx, odd);
  }
  return 0;
}
