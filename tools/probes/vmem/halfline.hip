// Do 64-byte (half-line) segments cost more than 128-byte ones in the GDFN tensors' access pattern?  (round 4)
// uir_gdfn_fwd writes t1 / t2 [pixels][192 floats] one 16 + 16 channel chunk at a time: per pixel and chunk two 64-byte segments
// (x1 half at column 16 c, x2 half at 96 + 16 c); uir_dwln_bwd / the gate-backward GEMM read them the same way.  An interleaved
// layout would make every (pixel, chunk) access one 128-byte line.  This probe streams 16 x 256 x 256 pixels x 192 floats through
// both patterns, chunk-outer inside 192-pixel tiles like the kernels, 16 bytes per lane, 8 lanes per pixel.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));

template <bool WRITE, bool IL, bool NT>
__global__ __launch_bounds__(256, 2) void k(float* __restrict__ buf, long npix, float* sink) {
  const int tid = threadIdx.x;
  const long tile0 = (long)blockIdx.x * 192;
  f4 acc = {0.f, 0.f, 0.f, 0.f};
  for (int c = 0; c < 6; ++c) {
#pragma unroll
    for (int it = 0; it < 6; ++it) {
      const int e = tid + 256 * it, p = e >> 3, j = e & 7;
      const long px = tile0 + p;
      if (px >= npix) continue;
      const int col = IL ? 32 * c + 4 * j : (j < 4 ? 16 * c + 4 * j : 96 + 16 * c + 4 * (j - 4));
      f4* ptr = reinterpret_cast<f4*>(buf + px * 192 + col);
      if (WRITE) {
        const f4 v = {(float)c, (float)j, 1.f, 2.f};
        if (NT) __builtin_nontemporal_store(v, ptr); else *ptr = v;
      } else {
        acc += NT ? __builtin_nontemporal_load(ptr) : *ptr;
      }
    }
    __syncthreads();
  }
  if (!WRITE && acc.x == 123.25f) sink[0] = acc.y;
}

template <bool WRITE, bool IL, bool NT>
void run(const char* name, float* buf, long npix, float* sink) {
  const int grid = (int)((npix + 191) / 192);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((k<WRITE, IL, NT>), dim3(grid), dim3(256), 0, 0, buf, npix, sink);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  for (int r = 0; r < 10; ++r) hipLaunchKernelGGL((k<WRITE, IL, NT>), dim3(grid), dim3(256), 0, 0, buf, npix, sink);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms;
  (void)hipEventElapsedTime(&ms, e0, e1);
  ms /= 10;
  printf("%-48s %8.1f us  %5.2f TB/s\n", name, ms * 1e3, npix * 192.0 * 4 / ms / 1e9);
}

int main() {
  const long npix = 16L * 256 * 256;
  float *buf, *sink;
  (void)hipMalloc(&buf, npix * 192 * 4);
  (void)hipMalloc(&sink, 4);
  (void)hipMemset(buf, 0, npix * 192 * 4);
  run<true, false, true>("write nt, two 64-B segments per pixel-chunk", buf, npix, sink);
  run<true, true, true>("write nt, one 128-B line per pixel-chunk", buf, npix, sink);
  run<true, false, false>("write,    two 64-B segments per pixel-chunk", buf, npix, sink);
  run<true, true, false>("write,    one 128-B line per pixel-chunk", buf, npix, sink);
  run<false, false, false>("read,     two 64-B segments per pixel-chunk", buf, npix, sink);
  run<false, true, false>("read,     one 128-B line per pixel-chunk", buf, npix, sink);
  run<false, false, true>("read nt,  two 64-B segments per pixel-chunk", buf, npix, sink);
  run<false, true, true>("read nt,  one 128-B line per pixel-chunk", buf, npix, sink);
  return 0;
}
