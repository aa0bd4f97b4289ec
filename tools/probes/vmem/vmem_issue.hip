// What does one vector-memory wave-instruction cost a CU?  (round 4: the tile-fused kernels spend 20 % of their time ISSUING small loads.)
// Every wave of a 256-thread workgroup (2 workgroups per CU, as uir_dwln_bwd runs) issues N independent loads of one shape from a small
// cache-resident table and sums them.  Shapes: dword with 64 consecutive lanes (256 B), dword as 4 rows x 64 B (the XP pattern), dword as
// 2 rows x 128 B, dwordx4 contiguous (1 KB), dwordx4 as 16 rows x 64 B.  Prints cycles per wave-instruction per CU (8 waves issuing).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int SHAPE>
__global__ __launch_bounds__(256, 2) void k(const float* __restrict__ tab, float* out, int iters, int stride) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float s = 0.f;
  // per-lane base offset (floats) of the shape; successive loads move by `stride` floats inside a 64 KB window (L1 / L2 hits)
  unsigned off;
  if (SHAPE == 0) off = lane;                                   // 256 B contiguous
  else if (SHAPE == 1) off = (lane >> 4) * 176 + (lane & 15);   // 4 rows x 64 B (rows 704 B apart)
  else if (SHAPE == 2) off = (lane >> 5) * 32 * 4 + (lane & 31);// 2 rows x 128 B
  else if (SHAPE == 3) off = lane * 4;                          // dwordx4, 1 KB contiguous
  else off = (lane >> 2) * 176 + (lane & 3) * 4;                // dwordx4, 16 rows x 64 B
  off += wave * 1024 + (blockIdx.x & 1) * 4096;
  for (int i = 0; i < iters; ++i) {
    float v[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const unsigned o = (off + (unsigned)(i * 16 + j) * stride) & 16383u;
      if (SHAPE < 3) v[j] = tab[o];
      else { const float4 t = *reinterpret_cast<const float4*>(tab + (o & ~3u)); v[j] = t.x + t.w; }
    }
#pragma unroll
    for (int j = 0; j < 16; ++j) s += v[j];
  }
  if (s == 123.456f) out[0] = s;
}

template <int SHAPE>
void run(const char* name, const float* tab, float* out) {
  const int iters = 2000, grid = 512;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<SHAPE>, dim3(grid), dim3(256), 0, 0, tab, out, 10, 52);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<SHAPE>, dim3(grid), dim3(256), 0, 0, tab, out, iters, 52);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double instr_per_cu = 8.0 * iters * 16;               // 8 waves per CU, one round of the grid (512 blocks on 256 CUs x 2)
  printf("%-34s %8.3f ms  %6.1f ns per wave-instruction per CU  (~%5.1f cycles at 2.1 GHz)\n", name, ms, ms * 1e6 / instr_per_cu, ms * 1e6 / instr_per_cu * 2.1);
}

int main() {
  float *tab, *out;
  hipMalloc(&tab, 65536 * 4 + 4096);
  hipMalloc(&out, 4);
  hipMemset(tab, 0, 65536 * 4 + 4096);
  run<0>("dword, 256 B contiguous", tab, out);
  run<1>("dword, 4 rows x 64 B", tab, out);
  run<2>("dword, 2 rows x 128 B", tab, out);
  run<3>("dwordx4, 1 KB contiguous", tab, out);
  run<4>("dwordx4, 16 rows x 64 B", tab, out);
  return 0;
}
