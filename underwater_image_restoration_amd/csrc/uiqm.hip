// UIQM on the GPU: getUIQM of /root/reference uqim_utils.py:176-190 for a batch of images of one size.
//   UIQM = 0.0282 UICM + 0.2953 UISM + 3.5753 UIConM
// The reference is numpy / scipy with Python pixel loops (seconds per 256 x 256 image).  Here:
//   uiqm_sobel_max   per (image, channel): max of the Sobel gradient magnitude (scipy.ndimage.sobel on both axes, 'reflect'
//                    border, np.hypot) - an order-independent max, folded with integer atomicMax on the float bits
//   uiqm_blocks      one wave per 10 x 10 window: the window's max / min of the three edge maps (magnitude * 255/max * channel,
//                    :50-54,97-99) and of the raw pixels over all channels -> the EME terms log(max/min) (:57-86) and the
//                    UIConM terms (top/bot) log(top/bot) (:160-173), in double like the reference's Python floats
//   uiqm_trimmed     one workgroup per (image, RG | YB plane): the asymmetric alpha-trimmed mean (:10-28) WITHOUT sorting -
//                    two 4-pass radix selects on order-preserving float keys find the first and the last kept value, one
//                    pass sums what lies strictly between and counts the ties at both ends, one pass gives the variance
//                    about the mean (:30-34)
//   uiqm_finalize    per image: fixed-order sums of the window terms, UICM / UISM / UIConM / UIQM
// Arithmetic follows the reference type by type (float32 where numpy computes in float32, double where Python floats do):
// explicit round-to-nearest intrinsics keep the compiler from contracting into FMAs.  Sums are taken in double in a fixed
// order; the reference's left-to-right float32 sum of the sorted values is exact for 8-bit images (every partial sum is a
// half-integer below 2^23), so results agree with it to double rounding there and to ~1e-5 for float-valued images.
// A window whose max/min ratio is <= 0 (signed input) sets flags[n]: the reference raises ValueError (math.log) - so does the host.
#include "uir_common.h"
#include "../../include/uir_kernels.h"

namespace {

constexpr int WIN = 10;

template <bool U8>
__device__ __forceinline__ float px(const UirUiqmArgs& a, long base, int y, int x, int c) {
  const long i = base + ((long)y * a.W + x) * 3 + c;
  if (U8) return (float)a.img_u8[i];
  return a.img_f32[i];
}

__device__ __forceinline__ int reflect(int i, int n) { return i < 0 ? 0 : (i >= n ? n - 1 : i); }   // scipy 'reflect': edge repeated

// Sobel magnitude of channel c at (y, x): every 1-D pass in double, stored as float32 (scipy line buffers); np.hypot in float32
// (glibc: sqrt of the double sum of squares, rounded once to float32)
template <bool U8>
__device__ __forceinline__ float sobel_mag(const UirUiqmArgs& a, long base, int y, int x, int c) {
  const int ym = reflect(y - 1, a.H), yp = reflect(y + 1, a.H), xm = reflect(x - 1, a.W), xp = reflect(x + 1, a.W);
  const float p00 = px<U8>(a, base, ym, xm, c), p01 = px<U8>(a, base, ym, x, c), p02 = px<U8>(a, base, ym, xp, c);
  const float p10 = px<U8>(a, base, y, xm, c), p12 = px<U8>(a, base, y, xp, c);
  const float p20 = px<U8>(a, base, yp, xm, c), p21 = px<U8>(a, base, yp, x, c), p22 = px<U8>(a, base, yp, xp, c);
  // axis 0 (rows): derivative down the column, then [1 2 1] along the row
  const float a0 = __fsub_rn(p20, p00), a1 = __fsub_rn(p21, p01), a2 = __fsub_rn(p22, p02);
  const float dx = (float)__dadd_rn(__dadd_rn((double)a0, __dmul_rn(2.0, (double)a1)), (double)a2);
  // axis 1 (columns): derivative along the row, then [1 2 1] down the column
  const float b0 = __fsub_rn(p02, p00), b1 = __fsub_rn(p12, p10), b2 = __fsub_rn(p22, p20);
  const float dy = (float)__dadd_rn(__dadd_rn((double)b0, __dmul_rn(2.0, (double)b1)), (double)b2);
  return (float)__dsqrt_rn(__dadd_rn(__dmul_rn((double)dx, (double)dx), __dmul_rn((double)dy, (double)dy)));
}

template <bool U8>
__global__ __launch_bounds__(256) void uiqm_sobel_max_kernel(const UirUiqmArgs a, unsigned* __restrict__ maxbits) {
  __shared__ float red[3][4];
  const int n = blockIdx.y;
  const long base = (long)n * a.H * a.W * 3;
  const long npx = (long)a.H * a.W;
  float m[3] = {0.f, 0.f, 0.f};
  for (long p = (long)blockIdx.x * 256 + threadIdx.x; p < npx; p += (long)gridDim.x * 256) {
    const int y = (int)(p / a.W), x = (int)(p - (long)y * a.W);
#pragma unroll
    for (int c = 0; c < 3; ++c) m[c] = fmaxf(m[c], sobel_mag<U8>(a, base, y, x, c));
  }
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    float v = m[c];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    if ((threadIdx.x & 63) == 0) red[c][threadIdx.x >> 6] = v;
  }
  __syncthreads();
  // ONE atomic per workgroup and channel (a wave-level atomic per 64 pixels serialised ~3000 same-address atomics per image:
  // 1.19 ms of the 1.96 ms batch).  Magnitudes are >= 0: the bit pattern orders like the value; max is order-independent.
  if (threadIdx.x < 3)
    atomicMax(maxbits + n * 3 + threadIdx.x, __float_as_uint(fmaxf(fmaxf(red[threadIdx.x][0], red[threadIdx.x][1]),
                                                                      fmaxf(red[threadIdx.x][2], red[threadIdx.x][3]))));
}

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  return v;
}
__device__ __forceinline__ float wave_min(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o));
  return v;
}
__device__ __forceinline__ int wave_or(int v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v |= __shfl_xor(v, o);
  return v;
}

// terms[n][4][k2*k1]: rows 0..2 = EME term of channel c, row 3 = UIConM term; window index = k * k1 + l
template <bool U8>
__global__ __launch_bounds__(256) void uiqm_blocks_kernel(const UirUiqmArgs a, const unsigned* __restrict__ maxbits,
                                                          double* __restrict__ terms, int k1, int k2) {
  const int n = blockIdx.y, lane = threadIdx.x & 63;
  const int blk = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (blk >= k1 * k2) return;
  const int k = blk / k1, l = blk - k * k1;
  const long base = (long)n * a.H * a.W * 3;
  float scale[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) scale[c] = __fdiv_rn(255.0f, __uint_as_float(maxbits[n * 3 + c]));   // 255.0 / np.max(mag), float32
  const float inf = __builtin_huge_valf();
  float emx[3] = {-inf, -inf, -inf}, emn[3] = {inf, inf, inf}, rmx = -inf, rmn = inf;
  int nan_e[3] = {0, 0, 0};
  for (int e = lane; e < WIN * WIN; e += 64) {
    const int y = k * WIN + e / WIN, x = l * WIN + e % WIN;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float v = px<U8>(a, base, y, x, c);
      rmx = fmaxf(rmx, v);
      rmn = fminf(rmn, v);
      const float ed = __fmul_rn(__fmul_rn(sobel_mag<U8>(a, base, y, x, c), scale[c]), v);   // (mag * 255/max) * channel
      nan_e[c] |= (ed != ed);                       // np.max / np.min propagate NaN (constant channel: 0 * inf)
      emx[c] = fmaxf(emx[c], ed);
      emn[c] = fminf(emn[c], ed);
    }
  }
  rmx = wave_max(rmx);
  rmn = wave_min(rmn);
  int bad = 0;
  const long nblk = (long)k1 * k2;
  double* t = terms + (long)n * 4 * nblk + blk;
  const double qnan = __longlong_as_double(0x7ff8000000000000LL);
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const float mx = wave_max(emx[c]), mn = wave_min(emn[c]);
    const int isnan_ = wave_or(nan_e[c]);
    if (lane == 0) {
      double term;
      if (isnan_) term = qnan;                                   // max_ = min_ = nan: neither == 0 test fires, log(nan) = nan
      else if (mn == 0.0f || mx == 0.0f) term = 0.0;
      else {
        const float ratio = __fdiv_rn(mx, mn);                   // np.float32 / np.float32
        if (!(ratio > 0.0f)) { bad = 1; term = 0.0; }            // math.log raises ValueError
        else term = log((double)ratio);
      }
      t[(long)c * nblk] = term;
    }
  }
  if (lane == 0) {
    const float top = __fsub_rn(rmx, rmn), bot = __fadd_rn(rmx, rmn);
    double term;
    if (top != top || bot != bot || bot == 0.0f || top == 0.0f) term = 0.0;
    else {
      const float ratio = __fdiv_rn(top, bot);
      if (!(ratio > 0.0f)) { bad = 1; term = 0.0; }
      else term = __dmul_rn((double)ratio, log((double)ratio));   // alpha * pow(r, alpha) * log(r), alpha = 1
    }
    t[3 * nblk] = term;
    if (bad) atomicOr(a.flags + n, 1);
  }
}

// ------------------------------------------------------------------------------------------------ UICM: trimmed means
__device__ __forceinline__ unsigned key_of(float v) {          // order-preserving: a < b  <=>  key(a) < key(b)
  const unsigned b = __float_as_uint(v);
  return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float val_of(unsigned k) {
  return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k);
}

template <bool U8>
__device__ __forceinline__ float plane_val(const UirUiqmArgs& a, long base, long p, int plane) {
  const long i = base + p * 3;
  float R, G, B;
  if (U8) { R = (float)a.img_u8[i]; G = (float)a.img_u8[i + 1]; B = (float)a.img_u8[i + 2]; }
  else { R = a.img_f32[i]; G = a.img_f32[i + 1]; B = a.img_f32[i + 2]; }
  if (plane == 0) return __fsub_rn(R, G);                                      // RG = R - G
  return __fsub_rn(__fdiv_rn(__fadd_rn(R, G), 2.0f), B);                        // YB = ((R + G) / 2) - B
}

constexpr int TT = 1024;   // threads of the trimmed-mean workgroup

__device__ double block_sum(double v, double* red) {              // fixed-order tree over the 1024 threads
  const int tid = threadIdx.x;
  red[tid] = v;
  __syncthreads();
  for (int s = TT / 2; s > 0; s >>= 1) {
    if (tid < s) red[tid] += red[tid + s];
    __syncthreads();
  }
  const double r = red[0];
  __syncthreads();
  return r;
}
__device__ long block_sum_l(long v, long* red) {
  const int tid = threadIdx.x;
  red[tid] = v;
  __syncthreads();
  for (int s = TT / 2; s > 0; s >>= 1) {
    if (tid < s) red[tid] += red[tid + s];
    __syncthreads();
  }
  const long r = red[0];
  __syncthreads();
  return r;
}

// stats[n][plane][2] = (mu_a as float32 widened, s_a)
template <bool U8>
__global__ __launch_bounds__(TT) void uiqm_trimmed_kernel(const UirUiqmArgs a, double* __restrict__ stats) {
  __shared__ unsigned hist[256];
  __shared__ double redd[TT];
  __shared__ unsigned sel_prefix;
  __shared__ long sel_rank;
  const int n = blockIdx.x >> 1, plane = blockIdx.x & 1, tid = threadIdx.x;
  const long K = (long)a.H * a.W, base = (long)n * K * 3;
  // T_a_L = ceil(0.1 K), T_a_R = floor(0.1 K) evaluated like Python: alpha * K in double
  const double aK = __dmul_rn(0.1, (double)K);
  const long tl = (long)ceil(aK), tr = (long)floor(aK);
  const long s = tl + 1, e = K - tr;                     // kept: sorted[s : e]
  unsigned kab[2] = {0u, 0u};
  if (e > s) {
    for (int which = 0; which < 2; ++which) {
      if (tid == 0) { sel_prefix = 0u; sel_rank = which == 0 ? s : e - 1; }
      __syncthreads();
      for (int pass = 0; pass < 4; ++pass) {
        const int shift = 24 - 8 * pass;
        if (tid < 256) hist[tid] = 0u;
        __syncthreads();
        const unsigned prefix = sel_prefix;
        const unsigned himask = pass == 0 ? 0u : (0xffffffffu << (shift + 8));
        for (long p = tid; p < K; p += TT) {
          const unsigned key = key_of(plane_val<U8>(a, base, p, plane));
          if ((key & himask) == prefix) atomicAdd(&hist[(key >> shift) & 255u], 1u);    // integer atomics: order-independent
        }
        __syncthreads();
        if (tid == 0) {
          long r = sel_rank;
          unsigned d = 0;
          for (; d < 255u; ++d) {
            if (r < (long)hist[d]) break;
            r -= hist[d];
          }
          sel_rank = r;
          sel_prefix = prefix | (d << shift);
        }
        __syncthreads();
      }
      kab[which] = sel_prefix;
      __syncthreads();
    }
  }
  const unsigned ka = kab[0], kb = kab[1];
  // sum of the values strictly between, ties counted at both ends
  double mid = 0.0;
  long less_a = 0, eq_a = 0, less_b = 0;
  if (e > s)
    for (long p = tid; p < K; p += TT) {
      const float v = plane_val<U8>(a, base, p, plane);
      const unsigned key = key_of(v);
      if (key > ka && key < kb) mid += (double)v;
      less_a += key < ka;
      eq_a += key == ka;
      less_b += key < kb;
    }
  mid = block_sum(mid, redd);
  long* redl = reinterpret_cast<long*>(redd);
  less_a = block_sum_l(less_a, redl);
  eq_a = block_sum_l(eq_a, redl);
  less_b = block_sum_l(less_b, redl);
  double total = 0.0;
  if (e > s) {
    const double va = (double)val_of(ka), vb = (double)val_of(kb);
    if (ka == kb) total = (double)(e - s) * va;
    else {
      const long na = (less_a + eq_a < e ? less_a + eq_a : e) - s;   // copies of the first kept value inside [s, e)
      const long nb = e - less_b;                                     // copies of the last kept value inside [s, e)
      total = (double)na * va + mid + (double)nb * vb;
    }
  }
  // weight * val: Python float times np.float32 -> float32 product of float32(weight) and the float32 sum
  const float weight = (float)(1.0 / (double)(K - tl - tr));
  const float mu = __fmul_rn(weight, (float)total);
  double sq = 0.0;
  for (long p = tid; p < K; p += TT) {
    const double d = (double)__fsub_rn(plane_val<U8>(a, base, p, plane), mu);     // (pixel - mu) in float32, squared in double
    sq += d * d;
  }
  sq = block_sum(sq, redd);
  if (tid == 0) {
    stats[(n * 2 + plane) * 2 + 0] = (double)mu;
    stats[(n * 2 + plane) * 2 + 1] = sq / (double)K;
  }
}

__global__ __launch_bounds__(256) void uiqm_finalize_kernel(const UirUiqmArgs a, const double* __restrict__ terms,
                                                            const double* __restrict__ stats, int k1, int k2) {
  __shared__ double red[256];
  const int n = blockIdx.x, tid = threadIdx.x;
  const long nblk = (long)k1 * k2;
  double sums[4];
  for (int row = 0; row < 4; ++row) {
    const double* t = terms + ((long)n * 4 + row) * nblk;
    double v = 0.0;
    for (long i = tid; i < nblk; i += 256) v += t[i];
    red[tid] = v;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
      if (tid < s) red[tid] += red[tid + s];
      __syncthreads();
    }
    sums[row] = red[0];
    __syncthreads();
  }
  if (tid == 0) {
    const double w_eme = 2.0 / (double)(k1 * k2), w_con = -1.0 / (double)(k1 * k2);
    const double uism = (0.299 * (w_eme * sums[0]) + 0.587 * (w_eme * sums[1])) + 0.144 * (w_eme * sums[2]);
    const double uiconm = w_con * sums[3];
    const double mu_rg = stats[(n * 2 + 0) * 2], s_rg = stats[(n * 2 + 0) * 2 + 1];
    const double mu_yb = stats[(n * 2 + 1) * 2], s_yb = stats[(n * 2 + 1) * 2 + 1];
    const double l = sqrt(mu_rg * mu_rg + mu_yb * mu_yb), r = sqrt(s_rg + s_yb);
    const double uicm = (-0.0268 * l) + (0.1586 * r);
    double* o = a.out + (long)n * 4;
    o[0] = ((0.0282 * uicm) + (0.2953 * uism)) + (3.5753 * uiconm);
    o[1] = uicm;
    o[2] = uism;
    o[3] = uiconm;
  }
}

template <bool U8>
int launch_all(const UirUiqmArgs& a, hipStream_t st) {
  const int k1 = a.W / WIN, k2 = a.H / WIN;
  const long nblk = (long)k1 * k2;
  unsigned* maxbits = reinterpret_cast<unsigned*>(a.work);                 // [N][3] (+ pad to a double boundary)
  double* terms = a.work + (a.N * 3 + 1) / 2 + 1;                          // [N][4][nblk]
  double* stats = terms + (long)a.N * 4 * nblk;                            // [N][2][2]
  if (hipMemsetAsync(maxbits, 0, sizeof(unsigned) * a.N * 3, st) != hipSuccess) return UIR_ELAUNCH;
  if (hipMemsetAsync(a.flags, 0, sizeof(int) * a.N, st) != hipSuccess) return UIR_ELAUNCH;
  const long npx = (long)a.H * a.W;
  int gx = (int)((npx + 2047) / 2048);        // ~8 pixels per thread: few same-address atomics, still >= 4 waves per CU at N >= 8
  if (gx > 512) gx = 512;
  hipLaunchKernelGGL((uiqm_sobel_max_kernel<U8>), dim3(gx, a.N), dim3(256), 0, st, a, maxbits);
  hipLaunchKernelGGL((uiqm_trimmed_kernel<U8>), dim3(2 * a.N), dim3(TT), 0, st, a, stats);
  hipLaunchKernelGGL((uiqm_blocks_kernel<U8>), dim3((unsigned)((nblk + 3) / 4), a.N), dim3(256), 0, st, a, maxbits, terms, k1, k2);
  hipLaunchKernelGGL(uiqm_finalize_kernel, dim3(a.N), dim3(256), 0, st, a, terms, stats, k1, k2);
  return uir_launch_status();
}

}  // namespace

extern "C" long uir_uiqm_work_doubles(int N, int H, int W) {
  if (N <= 0 || H < WIN || W < WIN) return -1;
  const long nblk = (long)(W / WIN) * (H / WIN);
  return (N * 3 + 1) / 2 + 1 + (long)N * 4 * nblk + (long)N * 4;
}

extern "C" int uir_uiqm(const UirUiqmArgs* p, void* stream) {
  const UirUiqmArgs& a = *p;
  UIR_REQUIRE((a.img_f32 != nullptr) != (a.img_u8 != nullptr));
  UIR_REQUIRE(a.work && a.out && a.flags && a.N > 0 && a.H >= WIN && a.W >= WIN);
  UIR_REQUIRE((long)a.N * a.H * a.W * 3 < (1L << 40) && (long)a.H * a.W < (1L << 31));
  hipStream_t st = static_cast<hipStream_t>(stream);
  return a.img_u8 ? launch_all<true>(a, st) : launch_all<false>(a, st);
}
