// Step tail of the reference trainer (src/ModelTrainer.py:87-88, :199): global-norm gradient
// clipping (clip_grad_norm_(params, 1.0)) and torch.optim.Adam defaults, fused over ONE flat
// fp32 buffer that holds every live parameter (and, under data parallelism, is the single RCCL
// all-reduce bucket).  The clip coefficient is computed on the device from the reduced sum of
// squares: no host synchronisation in the step.
#include "uir_common.h"
#include "../../include/uir_kernels.h"

namespace {

__global__ __launch_bounds__(256) void sumsq_partials_kernel(const float* __restrict__ g, long n, float* __restrict__ slab) {
  __shared__ float red[4];
  float s = 0.f;
  const long n4 = n >> 2;
  const float4* g4 = reinterpret_cast<const float4*>(g);
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    const float4 v = g4[i];
    s += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
  }
  if (blockIdx.x == 0 && threadIdx.x == 0)
    for (long i = n4 << 2; i < n; ++i) s += g[i] * g[i];
  s = uir_wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) slab[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

// Per-step scalars for a captured hipGraph: written by VALUE (kernel arguments are copied at launch time, so the host may
// run any number of steps ahead), read by the replayed clip_adam_kernel through `hyper`.  No host staging buffer exists.
__global__ void adam_hyper_kernel(float* hyper, float lr, float bc1, float bc2, float gscale) {
  if (threadIdx.x == 0) { hyper[0] = lr; hyper[1] = bc1; hyper[2] = bc2; hyper[3] = gscale; }
}

__global__ __launch_bounds__(256) void clip_adam_kernel(UirAdamArgs a) {
  if (a.hyper) {  // hipGraph replay only: the values uir_adam_set_hyper stored on this stream before the replay
    a.lr = a.hyper[0]; a.bc1 = a.hyper[1]; a.bc2 = a.hyper[2]; a.gscale = a.hyper[3];
  }
  const float total = sqrtf(a.sumsq[0]) * a.gscale;
  if (blockIdx.x == 0 && threadIdx.x == 0 && a.norm_out) a.norm_out[0] = total;
  float coef = a.max_norm / (total + 1e-6f);
  coef = (a.max_norm > 0.f) ? fminf(coef, 1.0f) : 1.0f;
  const float gmul = a.gscale * coef;
  const float step = a.lr / a.bc1;
  const float rs2 = rsqrtf(a.bc2);
  const float decay = 1.f - a.lr * a.weight_decay;   // AdamW (decoupled): p *= 1 - lr * wd before the update; 1 for Adam
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < a.n; i += (long)gridDim.x * blockDim.x) {
    const float g = a.g[i] * gmul;
    const float m = a.beta1 * a.m[i] + (1.f - a.beta1) * g;
    const float v = a.beta2 * a.v[i] + (1.f - a.beta2) * g * g;
    a.m[i] = m;
    a.v[i] = v;
    a.p[i] = a.p[i] * decay - step * m / (sqrtf(v) * rs2 + a.eps);
  }
}

}  // namespace

extern "C" int uir_sumsq_partials(const float* g, long n, float* slab, int nblocks, void* stream) {
  UIR_REQUIRE(g && slab && n > 0 && nblocks > 0 && uir_aligned16(g));
  hipLaunchKernelGGL(sumsq_partials_kernel, dim3(nblocks), dim3(256), 0, static_cast<hipStream_t>(stream), g, n, slab);
  return uir_launch_status();
}

extern "C" int uir_clip_adam_step(const UirAdamArgs* p, void* stream) {
  const UirAdamArgs& a = *p;
  UIR_REQUIRE(a.p && a.g && a.m && a.v && a.sumsq && a.n > 0);
  UIR_REQUIRE(a.hyper || (a.bc1 > 0.f && a.bc2 > 0.f));
  UIR_REQUIRE(a.weight_decay >= 0.f);
  long blocks = (a.n + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(clip_adam_kernel, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream), a);
  return uir_launch_status();
}

extern "C" int uir_adam_set_hyper(float* hyper, float lr, float bc1, float bc2, float gscale, void* stream) {
  UIR_REQUIRE(hyper && bc1 > 0.f && bc2 > 0.f);
  hipLaunchKernelGGL(adam_hyper_kernel, dim3(1), dim3(64), 0, static_cast<hipStream_t>(stream), hyper, lr, bc1, bc2, gscale);
  return uir_launch_status();
}

// A HIP stream whose kernels may only occupy `n_enabled` of the device's compute units (bits spread over the XCDs by the
// driver's symmetric mapping).  Used for the weight-gradient ("bulk") lane: its HBM-bound Gram kernels keep filling the
// bandwidth the latency-bound deep-level chain kernels leave idle, without taking every CU from them.
extern "C" int uir_stream_create_cu_masked(int n_enabled, void** out) {
  UIR_REQUIRE(out && n_enabled > 0);
  int dev = 0;
  hipDeviceProp_t prop;
  if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return UIR_ELAUNCH;
  const int total = prop.multiProcessorCount;
  if (n_enabled > total) n_enabled = total;
  uint32_t mask[16] = {0};
  UIR_REQUIRE(total <= 512);
  for (int i = 0; i < n_enabled; ++i) mask[i >> 5] |= 1u << (i & 31);
  hipStream_t st = nullptr;
  if (hipExtStreamCreateWithCUMask(&st, (uint32_t)((total + 31) / 32), mask) != hipSuccess) return UIR_ELAUNCH;
  *out = st;
  return UIR_OK;
}

extern "C" int uir_event_create(void** out) {
  UIR_REQUIRE(out);
  hipEvent_t ev = nullptr;
  if (hipEventCreateWithFlags(&ev, hipEventDisableTiming | hipEventDisableSystemFence) != hipSuccess) return UIR_ELAUNCH;
  *out = ev;
  return UIR_OK;
}
extern "C" int uir_event_destroy(void* ev) {
  UIR_REQUIRE(ev);
  return hipEventDestroy(static_cast<hipEvent_t>(ev)) == hipSuccess ? UIR_OK : UIR_ELAUNCH;
}
extern "C" int uir_event_record(void* ev, void* stream) {
  UIR_REQUIRE(ev);
  return hipEventRecord(static_cast<hipEvent_t>(ev), static_cast<hipStream_t>(stream)) == hipSuccess ? UIR_OK : UIR_ELAUNCH;
}
extern "C" int uir_stream_wait_event(void* stream, void* ev) {
  UIR_REQUIRE(ev);
  return hipStreamWaitEvent(static_cast<hipStream_t>(stream), static_cast<hipEvent_t>(ev), 0) == hipSuccess ? UIR_OK : UIR_ELAUNCH;
}
extern "C" int uir_stream_destroy(void* stream) {
  UIR_REQUIRE(stream);
  return hipStreamDestroy(static_cast<hipStream_t>(stream)) == hipSuccess ? UIR_OK : UIR_ELAUNCH;
}

extern "C" const char* uir_version(void) { return "uir_hip 0.1 gfx950"; }
extern "C" const char* uir_last_launch_error(void) { return hipGetErrorString(uir_last_hip_error); }
