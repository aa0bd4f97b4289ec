// The FIRST depthwise weight-gradient kernel (round 1-2: uir_dwconv_wgrad_kernel of csrc/dwconv.hip), archived here as the
// victim of the interference probe (tools/probes/pkfma/README.md).  It is no longer part of libuir_hip.so.
// hipcc's SLP vectoriser packs its 36 x 4 accumulator FMAs per row into v_pk_fma_f32; exactly two of the 72 packed FMAs of the row
// loop come out in the half-swapped form `op_sel:[0,1,0] op_sel_hi:[0,0,1]` (low result = src0.lo * src1.HI): the centre tap's x and z
// components, the only ones that were ever lost next to the split-bf16 contraction.
#include <hip/hip_runtime.h>

struct UirDwWgradArgs {          // layout of include/uir_kernels.h at the time
  const float* dY; const float* X; float* slab;
  int ldy, ldx, B, H, Wd, C, grp_real, grp_pad, creal, nblocks;
};

namespace {

constexpr int R = 4;

__device__ __forceinline__ int uir_colmap(int c, int real, int pad) {
  if (pad <= 0) return c;
  const int g = c / pad, o = c - g * pad;
  return o < real ? g * real + o : -1;
}

__device__ __forceinline__ float4 f4_fma(float4 a, float4 b, float4 c) {
  return make_float4(fmaf(a.x, b.x, c.x), fmaf(a.y, b.y, c.y), fmaf(a.z, b.z, c.z), fmaf(a.w, b.w, c.w));
}

__device__ __forceinline__ void load_row(float4 (&v)[R + 2], const float* base, long ld, int b, int hh, int w0, int H, int Wd) {
  const bool hok = hh >= 0 && hh < H;
  const int hc = min(max(hh, 0), H - 1);
  const float* rowp = base + ((long)(b * H + hc) * Wd) * ld;
#pragma unroll
  for (int j = 0; j < R + 2; ++j) {
    const int wc = min(max(w0 - 1 + j, 0), Wd - 1);
    v[j] = *reinterpret_cast<const float4*>(rowp + (long)wc * ld);
  }
#pragma unroll
  for (int j = 0; j < R + 2; ++j) {
    const int ww = w0 - 1 + j;
    if (!(hok && ww >= 0 && ww < Wd)) v[j] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
}

}  // namespace

extern "C" __global__ __launch_bounds__(256) void old_dwconv_wgrad_kernel(const UirDwWgradArgs a, const int HS) {
  extern __shared__ __attribute__((aligned(16))) float red[];  // [ny][c4n][36]
  const int c4n = a.C >> 2;
  const int c4 = threadIdx.x, py = threadIdx.y, ny = blockDim.y;
  const int runs = (a.Wd + R - 1) / R;
  const int strips = (a.H + HS - 1) / HS;
  const long items = (long)a.B * strips * runs;
  float4 acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) acc[t] = make_float4(0.f, 0.f, 0.f, 0.f);
  if (c4 < c4n) {
    const float* xb = a.X + c4 * 4;
    for (long it = (long)blockIdx.x * ny + py; it < items; it += (long)gridDim.x * ny) {
      const int run = (int)(it % runs);
      long t = it / runs;
      const int sp = (int)(t % strips);
      const int b = (int)(t / strips);
      const int w0 = run * R, h0 = sp * HS;
      const int hend = min(h0 + HS, a.H);
      float4 r0[R + 2], r1[R + 2], r2[R + 2];
      load_row(r0, xb, a.ldx, b, h0 - 1, w0, a.H, a.Wd);
      load_row(r1, xb, a.ldx, b, h0, w0, a.H, a.Wd);
      for (int h = h0; h < hend; ++h) {
        load_row(r2, xb, a.ldx, b, h + 1, w0, a.H, a.Wd);
        float4 g[R];
        const float* gp = a.dY + ((long)(b * a.H + h) * a.Wd) * a.ldy + c4 * 4;
#pragma unroll
        for (int j = 0; j < R; ++j) g[j] = *reinterpret_cast<const float4*>(gp + (long)min(w0 + j, a.Wd - 1) * a.ldy);
#pragma unroll
        for (int j = 0; j < R; ++j)
          if (w0 + j >= a.Wd) g[j] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int dx = 0; dx < 3; ++dx)
#pragma unroll
          for (int j = 0; j < R; ++j) {
            acc[dx] = f4_fma(g[j], r0[j + dx], acc[dx]);
            acc[3 + dx] = f4_fma(g[j], r1[j + dx], acc[3 + dx]);
            acc[6 + dx] = f4_fma(g[j], r2[j + dx], acc[6 + dx]);
          }
#pragma unroll
        for (int j = 0; j < R + 2; ++j) { r0[j] = r1[j]; r1[j] = r2[j]; }
      }
    }
    float* rp = red + ((long)py * c4n + c4) * 36;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      rp[0 * 9 + t] = acc[t].x; rp[1 * 9 + t] = acc[t].y; rp[2 * 9 + t] = acc[t].z; rp[3 * 9 + t] = acc[t].w;
    }
  }
  __syncthreads();
  // slab[block][creal][9]: sum over py in fixed order
  const int tid = threadIdx.y * blockDim.x + threadIdx.x, nthr = blockDim.x * blockDim.y;
  for (int e = tid; e < a.C * 9; e += nthr) {
    const int c = e / 9, t = e - c * 9;
    const int cr = uir_colmap(c, a.grp_real, a.grp_pad);
    if (cr < 0) continue;
    float s = 0.f;
    for (int y = 0; y < ny; ++y) s += red[((long)y * c4n + (c >> 2)) * 36 + (c & 3) * 9 + t];
    a.slab[(long)blockIdx.x * a.creal * 9 + (long)cr * 9 + t] = s;
  }
}
