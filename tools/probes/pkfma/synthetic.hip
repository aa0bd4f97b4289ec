// Register-only victims for the interference probe: a long loop of ONE packed-f32 instruction form (inline asm), small exact
// integer operands, eight independent accumulator pairs.  Run alone and beside the split-bf16 contraction; any lost or altered
// result shows as a bitwise difference (run_synthetic.py).  Build: hipcc --offload-arch=gfx950 -O3 -shared -fPIC synthetic.hip -o libpkfma_synth.so
#include <hip/hip_runtime.h>

typedef float v2f __attribute__((ext_vector_type(2)));

#define FORM_LIST(X)                                                                                         \
  X(0, "v_pk_fma_f32 %0, %1, %2, %0")                                                                        \
  X(1, "v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]")                                                      \
  X(2, "v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[0,0,1]")                                       \
  X(3, "v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[0,1,1]")                                       \
  X(4, "v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[1,1,1]")                                       \
  X(5, "v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[1,0,1]")                                       \
  X(6, "v_pk_mul_f32 %1, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]\n\tv_pk_add_f32 %0, %0, %1")                    \
  X(7, "v_pk_add_f32 %0, %0, %2 op_sel:[0,1] op_sel_hi:[1,0]")

template <int FORM>
__device__ __forceinline__ void op(v2f& acc, v2f& a, const v2f& b);
#define X(N, TXT)                                                                          \
  template <>                                                                              \
  __device__ __forceinline__ void op<N>(v2f & acc, v2f & a, const v2f& b) {                \
    asm volatile(TXT : "+v"(acc), "+v"(a) : "v"(b));                                       \
  }
FORM_LIST(X)
#undef X

// form 6 overwrites `a` with the product: a.x, a.y in {-1, +1} and b in {-1, +1} keep every intermediate exact
template <int FORM>
__global__ __launch_bounds__(256) void victim(const float* in, float* out, int iters) {
  const int tid = blockIdx.x * blockDim.x + threadIdx.x;
  v2f a, b, acc[8];
  a.x = in[tid * 4 + 0]; a.y = in[tid * 4 + 1];
  b.x = in[tid * 4 + 2]; b.y = in[tid * 4 + 3];
#pragma unroll
  for (int i = 0; i < 8; ++i) { acc[i].x = 0.f; acc[i].y = 0.f; }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) op<FORM>(acc[i], a, b);
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    out[(tid * 8 + i) * 2 + 0] = acc[i].x;
    out[(tid * 8 + i) * 2 + 1] = acc[i].y;
  }
}

extern "C" int launch_victim(int form, const float* in, float* out, int iters, int blocks, void* stream) {
  hipStream_t st = static_cast<hipStream_t>(stream);
  switch (form) {
#define X(N, TXT) case N: hipLaunchKernelGGL(victim<N>, dim3(blocks), dim3(256), 0, st, in, out, iters); break;
    FORM_LIST(X)
#undef X
    default: return 1;
  }
  return hipGetLastError() == hipSuccess ? 0 : 2;
}

extern "C" const char* form_text(int form) {
  switch (form) {
#define X(N, TXT) case N: return TXT;
    FORM_LIST(X)
#undef X
  }
  return "";
}
