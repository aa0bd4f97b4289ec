// Phase boundaries of the tile-fused kernels.  UIR_ST(i) marks the end of phase i: in the product build it is a scheduling fence
// (see the bottom of this file); in the DIAGNOSTIC builds (tools/probes/stamps/: -DUIR_STAMPS=1) it also stamps the phase time.  Each wave keeps per-phase cycle sums in SGPRs (s_memtime deltas, no vector registers,
// no vector-memory traffic); lane 0 writes them to a buffer of the code object at the end.  MI355X_MICROARCH.md: a stamp costs an
// s_waitcnt lgkmcnt(0) - read the sums as a breakdown, not as absolute times.
#pragma once
#if defined(UIR_STAMPS) && UIR_STAMPS >= 2
// probes of what the stamps change besides measuring: 2 = a scheduling barrier at every stamp site, 3 = an LDS / scalar-memory drain
#define UIR_ST_DECL
#if UIR_STAMPS == 2
#define UIR_ST(i) __builtin_amdgcn_sched_barrier(0)
#else
#define UIR_ST(i) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
#endif
#define UIR_ST_FLUSH()
#define UIR_ST_EXPORT
#elif defined(UIR_STAMPS)
#define UIR_NST 12
__device__ unsigned uir_stamp_buf[1024 * 4 * UIR_NST];
#define UIR_ST_DECL unsigned st_prev_ = (unsigned)__builtin_amdgcn_s_memtime(); unsigned st_acc_[UIR_NST] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}
#define UIR_ST(i) do { const unsigned now_ = (unsigned)__builtin_amdgcn_s_memtime(); st_acc_[i] += now_ - st_prev_; st_prev_ = now_; } while (0)
#define UIR_ST_FLUSH() do { if ((threadIdx.x & 63) == 0 && blockIdx.x < 1024) { _Pragma("unroll") for (int i_ = 0; i_ < UIR_NST; ++i_) \
    uir_stamp_buf[(blockIdx.x * 4 + (threadIdx.x >> 6)) * UIR_NST + i_] = st_acc_[i_]; } } while (0)
#define UIR_ST_EXPORT extern "C" int uir_stamps_fetch(unsigned* host, int n) { \
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(uir_stamp_buf), (size_t)n * sizeof(unsigned)); }
#else
// Product build: a phase boundary is a SCHEDULING FENCE.  Without it hipcc moves work across the phases of a chunk (measured on
// uir_dwln_bwd<32, 6, true>, 16 x 256 x 256: 836 us as compiled, 719 us with the fences and nothing else changed; profiles/r04).
#define UIR_ST_DECL
#define UIR_ST(i) __builtin_amdgcn_sched_barrier(0)
#define UIR_ST_FLUSH()
#define UIR_ST_EXPORT
#endif
