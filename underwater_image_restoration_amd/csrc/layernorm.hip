// uir_layernorm_bwd: input-gradient of the per-pixel channel LayerNorm, fused with the
// residual-path gradient add, plus per-workgroup partials of dgamma / dbeta.
// A row (pixel) of C channels is held by L = C/4 adjacent lanes, one float4 each; the two row
// reductions are xor-shuffles inside those L lanes.  HBM-bound: reads dT, X (+dRes), writes dX.
#include "uir_common.h"
#include "../../include/uir_kernels.h"

namespace {

template <int L>
__global__ __launch_bounds__(256) void uir_ln_bwd_kernel(const UirLnBwdArgs a) {
  __shared__ float red[256 * 8];
  constexpr int ROWS = 256 / L;
  const int li = threadIdx.x % L;        // float4 index inside the row
  const int rl = threadIdx.x / L;        // row inside the block iteration
  const float invc = 1.0f / (float)a.C;
  const float4 gam = *reinterpret_cast<const float4*>(a.gamma + li * 4);
  float4 dg = make_float4(0.f, 0.f, 0.f, 0.f), db = make_float4(0.f, 0.f, 0.f, 0.f);
  for (long m0 = (long)blockIdx.x * ROWS; m0 < a.M; m0 += (long)gridDim.x * ROWS) {
    const long m = m0 + rl;
    const bool ok = m < a.M;
    float4 x = make_float4(0.f, 0.f, 0.f, 0.f), dt = x;
    float mean = 0.f, rstd = 0.f;
    if (ok) {
      x = *reinterpret_cast<const float4*>(a.X + m * a.ldx + li * 4);
      dt = *reinterpret_cast<const float4*>(a.dT + m * a.ldt + li * 4);
      mean = a.stats[m * 2];
      rstd = a.stats[m * 2 + 1];
    }
    const float4 xh = make_float4((x.x - mean) * rstd, (x.y - mean) * rstd, (x.z - mean) * rstd, (x.w - mean) * rstd);
    const float4 gd = make_float4(dt.x * gam.x, dt.y * gam.y, dt.z * gam.z, dt.w * gam.w);
    float s1 = (gd.x + gd.y) + (gd.z + gd.w);
    float s2 = (gd.x * xh.x + gd.y * xh.y) + (gd.z * xh.z + gd.w * xh.w);
#pragma unroll
    for (int o = L / 2; o > 0; o >>= 1) {
      s1 += __shfl_xor(s1, o);
      s2 += __shfl_xor(s2, o);
    }
    s1 *= invc;
    s2 *= invc;
    if (ok) {
      float4 dx = make_float4(rstd * (gd.x - s1 - xh.x * s2), rstd * (gd.y - s1 - xh.y * s2),
                              rstd * (gd.z - s1 - xh.z * s2), rstd * (gd.w - s1 - xh.w * s2));
      if (a.dRes) {
        const float4 rr = *reinterpret_cast<const float4*>(a.dRes + m * a.ldr + li * 4);
        dx.x += rr.x; dx.y += rr.y; dx.z += rr.z; dx.w += rr.w;
      }
      *reinterpret_cast<float4*>(a.dX + m * a.ldo + li * 4) = dx;
      dg.x += dt.x * xh.x; dg.y += dt.y * xh.y; dg.z += dt.z * xh.z; dg.w += dt.w * xh.w;
      db.x += dt.x; db.y += dt.y; db.z += dt.z; db.w += dt.w;
    }
  }
  float* rp = red + threadIdx.x * 8;
  rp[0] = dg.x; rp[1] = dg.y; rp[2] = dg.z; rp[3] = dg.w;
  rp[4] = db.x; rp[5] = db.y; rp[6] = db.z; rp[7] = db.w;
  __syncthreads();
  // slab[block][2C]: thread t < 2C sums its column over the ROWS row-slots (fixed order)
  if ((int)threadIdx.x < 2 * a.C) {
    const int which = threadIdx.x / a.C;      // 0 = dgamma, 1 = dbeta
    const int c = threadIdx.x - which * a.C;
    float s = 0.f;
    for (int rr = 0; rr < ROWS; ++rr) s += red[(rr * L + (c >> 2)) * 8 + which * 4 + (c & 3)];
    a.slab[(long)blockIdx.x * 2 * a.C + threadIdx.x] = s;
  }
}

// C = 256 / 512: one wave per row, VPL float4 per lane; dgamma / dbeta accumulate per lane over the wave's rows and the four
// waves are summed through LDS (fixed order) into the workgroup's slab row [dgamma(C) | dbeta(C)].
template <int VPL>
__global__ __launch_bounds__(256) void uir_ln_bwd_wide_kernel(const UirLnBwdArgs a) {
  __shared__ float red[4][2][256 * VPL];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float invc = 1.0f / (float)a.C;
  float4 gam[VPL], dg[VPL], db[VPL];
#pragma unroll
  for (int v = 0; v < VPL; ++v) {
    gam[v] = *reinterpret_cast<const float4*>(a.gamma + 4 * (lane + 64 * v));
    dg[v] = make_float4(0.f, 0.f, 0.f, 0.f);
    db[v] = dg[v];
  }
  for (long m = (long)blockIdx.x * 4 + wave; m < a.M; m += (long)gridDim.x * 4) {
    const float mean = a.stats[m * 2], rstd = a.stats[m * 2 + 1];
    float4 xh[VPL], gd[VPL], dt[VPL];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int v = 0; v < VPL; ++v) {
      const int c = 4 * (lane + 64 * v);
      const float4 x = *reinterpret_cast<const float4*>(a.X + m * a.ldx + c);
      dt[v] = *reinterpret_cast<const float4*>(a.dT + m * a.ldt + c);
      xh[v] = make_float4((x.x - mean) * rstd, (x.y - mean) * rstd, (x.z - mean) * rstd, (x.w - mean) * rstd);
      gd[v] = make_float4(dt[v].x * gam[v].x, dt[v].y * gam[v].y, dt[v].z * gam[v].z, dt[v].w * gam[v].w);
      s1 += (gd[v].x + gd[v].y) + (gd[v].z + gd[v].w);
      s2 += (gd[v].x * xh[v].x + gd[v].y * xh[v].y) + (gd[v].z * xh[v].z + gd[v].w * xh[v].w);
    }
    s1 = uir_wave_sum(s1) * invc;
    s2 = uir_wave_sum(s2) * invc;
#pragma unroll
    for (int v = 0; v < VPL; ++v) {
      const int c = 4 * (lane + 64 * v);
      float4 dx = make_float4(rstd * (gd[v].x - s1 - xh[v].x * s2), rstd * (gd[v].y - s1 - xh[v].y * s2),
                              rstd * (gd[v].z - s1 - xh[v].z * s2), rstd * (gd[v].w - s1 - xh[v].w * s2));
      if (a.dRes) {
        const float4 rr = *reinterpret_cast<const float4*>(a.dRes + m * a.ldr + c);
        dx.x += rr.x; dx.y += rr.y; dx.z += rr.z; dx.w += rr.w;
      }
      *reinterpret_cast<float4*>(a.dX + m * a.ldo + c) = dx;
      dg[v].x += dt[v].x * xh[v].x; dg[v].y += dt[v].y * xh[v].y; dg[v].z += dt[v].z * xh[v].z; dg[v].w += dt[v].w * xh[v].w;
      db[v].x += dt[v].x; db[v].y += dt[v].y; db[v].z += dt[v].z; db[v].w += dt[v].w;
    }
  }
#pragma unroll
  for (int v = 0; v < VPL; ++v) {
    const int c = 4 * (lane + 64 * v);
    *reinterpret_cast<float4*>(&red[wave][0][c]) = dg[v];
    *reinterpret_cast<float4*>(&red[wave][1][c]) = db[v];
  }
  __syncthreads();
  for (int t = threadIdx.x; t < 2 * a.C; t += 256) {
    const int which = t / a.C, c = t - which * a.C;
    a.slab[(long)blockIdx.x * 2 * a.C + t] = (red[0][which][c] + red[1][which][c]) + (red[2][which][c] + red[3][which][c]);
  }
}

}  // namespace

extern "C" int uir_layernorm_bwd(const UirLnBwdArgs* p, void* stream) {
  const UirLnBwdArgs& a = *p;
  UIR_REQUIRE(a.dT && a.X && a.stats && a.gamma && a.dX && a.slab);
  UIR_REQUIRE(a.M > 0 && a.nblocks > 0);
  UIR_REQUIRE(a.C == 16 || a.C == 32 || a.C == 64 || a.C == 128 || a.C == 256 || a.C == 512);
  UIR_REQUIRE((a.ldt & 3) == 0 && (a.ldx & 3) == 0 && (a.ldo & 3) == 0 && a.ldt >= a.C && a.ldx >= a.C && a.ldo >= a.C);
  UIR_REQUIRE(uir_aligned16(a.dT) && uir_aligned16(a.X) && uir_aligned16(a.dX) && uir_aligned16(a.gamma));
  if (a.dRes) UIR_REQUIRE((a.ldr & 3) == 0 && a.ldr >= a.C && uir_aligned16(a.dRes));
  hipStream_t st = static_cast<hipStream_t>(stream);
  dim3 grid(a.nblocks), block(256);
  switch (a.C) {
    case 16: hipLaunchKernelGGL(uir_ln_bwd_kernel<4>, grid, block, 0, st, a); break;
    case 32: hipLaunchKernelGGL(uir_ln_bwd_kernel<8>, grid, block, 0, st, a); break;
    case 64: hipLaunchKernelGGL(uir_ln_bwd_kernel<16>, grid, block, 0, st, a); break;
    case 128: hipLaunchKernelGGL(uir_ln_bwd_kernel<32>, grid, block, 0, st, a); break;
    case 256: hipLaunchKernelGGL(uir_ln_bwd_wide_kernel<1>, grid, block, 0, st, a); break;
    default: hipLaunchKernelGGL(uir_ln_bwd_wide_kernel<2>, grid, block, 0, st, a); break;
  }
  return uir_launch_status();
}
