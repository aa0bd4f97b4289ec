// Window attention core of WindowAttention_Sparse (reference src/model/block.py:284-370), forward and
// backward, for the 8 x 8 windows every DecoderBlock / MDASSA of NewBigFRFNModel uses (model.py:517-...: win_size 8,
// shift_size 0, mask None) and, with args.shift = 4, the shifted windows + region mask of AST's TransformerBlock
// (src/Models/AST.py:144-219,568-616: same attention arithmetic, scores + mask before softmax and relu):
//   attn = (q * scale) k^T + relative_position_bias            [heads, 64, 64] per window          (:340-352)
//   attn = softmax(attn) * w1 + relu(attn)^2 * w2,  (w1, w2) = softmax(self.w)                      (:361-367)
//   out  = attn v, heads concatenated along the channels                                             (:370)
// The Linear projections around it (to_q / to_kv / proj, :322,327,371) are uir_gemm calls on the token rows: a window is just a
// set of 64 rows of the HWC token image, gathered here by address - window_partition / window_reverse (:16-41) never move data.
// One 64-lane workgroup per (window, head); lane i owns query token i: its score row, softmax and output row live in registers,
// K / V (and, backward, Q / dO and the two 64 x 64 matrices P and dS) are staged in LDS.  relative_position_index (:302-311) is
// evaluated from the token coordinates.  Backward writes dQ / dK / dV rows and per-workgroup partials of the bias-table and
// mixing-weight gradients (summed in fixed order by uir_reduce_slab(_batch); no float atomics).
#include "uir_common.h"
#include "../../include/uir_kernels.h"

namespace {

constexpr int WS = 8, NT = WS * WS, RP = 2 * WS - 1;   // window side, tokens per window, relative positions per axis

__device__ __forceinline__ long token_row(const UirWinAttnArgs& a, int win, int t) {
  const int wpr = a.W / WS, wpi = (a.H / WS) * wpr;
  const int b = win / wpi, r = win - b * wpi;
  const int wy = r / wpr, wx = r - wy * wpr;
  int y = wy * WS + (t >> 3) + a.shift, x = wx * WS + (t & 7) + a.shift;     // shifted windows: the cyclic roll is an address
  y -= y >= a.H ? a.H : 0;
  x -= x >= a.W ? a.W : 0;
  return ((long)b * a.H + y) * a.W + x;
}

// Shift-region id of window token t (AST.py:571-582: three row bands x three column bands of the ROLLED image, cut at
// n - ws and n - shift); tokens of different regions get the additive -100 (:586-588).  shift = 0: one region.
__device__ __forceinline__ int shift_region(const UirWinAttnArgs& a, int win, int t) {
  const int wpr = a.W / WS, wpi = (a.H / WS) * wpr;
  const int r = win % wpi;
  const int y = (r / wpr) * WS + (t >> 3), x = (r % wpr) * WS + (t & 7);
  const int ry = y < a.H - WS ? 0 : (y < a.H - a.shift ? 1 : 2);
  const int rx = x < a.W - WS ? 0 : (x < a.W - a.shift ? 1 : 2);
  return ry * 3 + rx;
}

// One channel chunk (CH <= 32 floats of head h, starting at column c0) of the 64 window tokens -> LDS rows, unpadded and 16-byte
// aligned: the hot loops read them as broadcast ds_read_b128 (every lane the same address); only this one store conflicts.
template <int CH>
__device__ __forceinline__ void stage_chunk(const float* __restrict__ base, int ld, int c0, const UirWinAttnArgs& a, int win,
                                            float (*dst)[CH]) {
  const int t = threadIdx.x;
  const float* p = base + token_row(a, win, t) * ld + c0;
#pragma unroll
  for (int c = 0; c < CH; c += 4) *reinterpret_cast<float4*>(&dst[t][c]) = *reinterpret_cast<const float4*>(p + c);
}

template <int CH>
__device__ __forceinline__ void load_chunk(const float* __restrict__ row, float (&v)[CH]) {
#pragma unroll
  for (int c = 0; c < CH; c += 4) {
    const float4 t = *reinterpret_cast<const float4*>(row + c);
    v[c] = t.x; v[c + 1] = t.y; v[c + 2] = t.z; v[c + 3] = t.w;
  }
}

__device__ __forceinline__ void mix_weights(const float* w, float& w1, float& w2) {
  const float m = fmaxf(w[0], w[1]);
  const float e0 = __expf(w[0] - m), e1 = __expf(w[1] - m);
  w1 = e0 / (e0 + e1);
  w2 = e1 / (e0 + e1);
}

// scores of query token i against the 64 keys of its window: s = (q . k) scale + bias; p0 = softmax(s).  The head dimension is
// walked in chunks of CH channels (K chunk staged in LDS, the lane's q chunk in registers): any head dim from 8 to 128.
template <int HD, int CH>
__device__ __forceinline__ void score_row(const UirWinAttnArgs& a, int win, int h, int i, long row, float (*T)[CH], float (&s)[NT],
                                          float (&p0)[NT]) {
#pragma unroll
  for (int j = 0; j < NT; ++j) s[j] = 0.f;
  for (int c0 = 0; c0 < HD; c0 += CH) {
    stage_chunk<CH>(a.K, a.ldk, h * HD + c0, a, win, T);
    float q[CH];
    load_chunk<CH>(a.Q + row * a.ldq + h * HD + c0, q);
    __syncthreads();
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      float d = s[j];
#pragma unroll
      for (int c = 0; c < CH; ++c) d = fmaf(q[c], T[j][c], d);
      s[j] = d;
    }
    __syncthreads();
  }
  const int yi = i >> 3, xi = i & 7;
  float mx = -__builtin_huge_valf();
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int idx = (yi - (j >> 3) + WS - 1) * RP + (xi - (j & 7) + WS - 1);
    s[j] = fmaf(s[j], a.scale, a.table[idx * a.heads + h]);
  }
  if (a.shift > 0) {                 // wave-uniform branch; the 64 region ids of the window travel by shuffle
    const int mine = shift_region(a, win, i);
#pragma unroll
    for (int j = 0; j < NT; ++j) s[j] += (__shfl(mine, j) != mine) ? -100.0f : 0.0f;
  }
#pragma unroll
  for (int j = 0; j < NT; ++j) mx = fmaxf(mx, s[j]);
  float sum = 0.f;
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    p0[j] = __expf(s[j] - mx);
    sum += p0[j];
  }
  const float inv = 1.0f / sum;
#pragma unroll
  for (int j = 0; j < NT; ++j) p0[j] *= inv;
}

template <int HD>
__global__ __launch_bounds__(64) void winattn_fwd_kernel(const UirWinAttnArgs a) {
  constexpr int CH = HD < 32 ? HD : 32;
  __shared__ __align__(16) float T[NT][CH];
  const int win = blockIdx.x, h = blockIdx.y, i = threadIdx.x;
  const long row = token_row(a, win, i);
  float s[NT], p0[NT];
  score_row<HD, CH>(a, win, h, i, row, T, s, p0);
  float w1, w2;
  mix_weights(a.w, w1, w2);
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const float r = fmaxf(s[j], 0.f);
    p0[j] = fmaf(w1, p0[j], w2 * r * r);            // the mixed attention row
  }
  for (int c0 = 0; c0 < HD; c0 += CH) {
    stage_chunk<CH>(a.V, a.ldv, h * HD + c0, a, win, T);
    __syncthreads();
    float o[CH];
#pragma unroll
    for (int c = 0; c < CH; ++c) o[c] = 0.f;
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int c = 0; c < CH; ++c) o[c] = fmaf(p0[j], T[j][c], o[c]);
#pragma unroll
    for (int c = 0; c < CH; c += 4)
      *reinterpret_cast<float4*>(a.O + row * a.ldo + h * HD + c0 + c) = make_float4(o[c], o[c + 1], o[c + 2], o[c + 3]);
    __syncthreads();
  }
}

template <int HD>
__global__ __launch_bounds__(64) void winattn_bwd_kernel(const UirWinAttnArgs a) {
  constexpr int CH = HD < 32 ? HD : 32;
  __shared__ __align__(16) float T[NT][CH];
  __shared__ __align__(16) float T2[NT][CH];
  __shared__ float Pm[NT][NT + 1], Dm[NT][NT + 1];                // P and dS, row = query token
  const int win = blockIdx.x, h = blockIdx.y, i = threadIdx.x;
  const long row = token_row(a, win, i);
  float s[NT], p0[NT];
  score_row<HD, CH>(a, win, h, i, row, T, s, p0);
  float w1, w2;
  mix_weights(a.w, w1, w2);
  // dP_j = dO_i . v_j
  float dp[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j) dp[j] = 0.f;
  for (int c0 = 0; c0 < HD; c0 += CH) {
    stage_chunk<CH>(a.V, a.ldv, h * HD + c0, a, win, T);
    float g[CH];
    load_chunk<CH>(a.dO + row * a.ldo + h * HD + c0, g);
    __syncthreads();
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      float d = dp[j];
#pragma unroll
      for (int c = 0; c < CH; ++c) d = fmaf(g[c], T[j][c], d);
      dp[j] = d;
    }
    __syncthreads();
  }
  float sp = 0.f, dw2 = 0.f;
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const float r = fmaxf(s[j], 0.f);
    sp = fmaf(dp[j], p0[j], sp);
    dw2 = fmaf(dp[j], r * r, dw2);
  }
  float dw1 = sp;
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const float r = fmaxf(s[j], 0.f);
    const float ds = fmaf(w1 * p0[j], dp[j] - sp, w2 * dp[j] * 2.f * r);   // softmax backward + relu^2 backward
    Pm[i][j] = fmaf(w1, p0[j], w2 * r * r);
    Dm[i][j] = ds;
    dp[j] = ds;                                                             // keep dS_i in registers for dQ
  }
  // mixing-weight partials: sum over the 64 query rows of this (window, head)
  dw1 = uir_wave_sum(dw1);
  dw2 = uir_wave_sum(dw2);
  if (i == 0) {
    float* o = a.dw_slab + ((long)win * a.heads + h) * 2;
    o[0] = dw1;
    o[1] = dw2;
  }
  // dQ_i = scale * sum_j dS_ij k_j
  for (int c0 = 0; c0 < HD; c0 += CH) {
    stage_chunk<CH>(a.K, a.ldk, h * HD + c0, a, win, T);
    __syncthreads();
    float dq[CH];
#pragma unroll
    for (int c = 0; c < CH; ++c) dq[c] = 0.f;
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int c = 0; c < CH; ++c) dq[c] = fmaf(dp[j], T[j][c], dq[c]);
#pragma unroll
    for (int c = 0; c < CH; c += 4)
      *reinterpret_cast<float4*>(a.dQ + row * a.lddq + h * HD + c0 + c) =
          make_float4(dq[c] * a.scale, dq[c + 1] * a.scale, dq[c + 2] * a.scale, dq[c + 3] * a.scale);
    __syncthreads();
  }
  // lane j: dK_j = scale * sum_i dS[i][j] q_i ;  dV_j = sum_i P[i][j] dO_i   (Pm / Dm were completed before the barriers above)
  for (int c0 = 0; c0 < HD; c0 += CH) {
    stage_chunk<CH>(a.Q, a.ldq, h * HD + c0, a, win, T);
    stage_chunk<CH>(a.dO, a.ldo, h * HD + c0, a, win, T2);
    __syncthreads();
    float dk[CH], dv[CH];
#pragma unroll
    for (int c = 0; c < CH; ++c) { dk[c] = 0.f; dv[c] = 0.f; }
#pragma unroll 4
    for (int r = 0; r < NT; ++r) {
      const float ds = Dm[r][i], p = Pm[r][i];
#pragma unroll
      for (int c = 0; c < CH; ++c) {
        dk[c] = fmaf(ds, T[r][c], dk[c]);
        dv[c] = fmaf(p, T2[r][c], dv[c]);
      }
    }
#pragma unroll
    for (int c = 0; c < CH; c += 4) {
      *reinterpret_cast<float4*>(a.dK + row * a.lddk + h * HD + c0 + c) =
          make_float4(dk[c] * a.scale, dk[c + 1] * a.scale, dk[c + 2] * a.scale, dk[c + 3] * a.scale);
      *reinterpret_cast<float4*>(a.dV + row * a.lddv + h * HD + c0 + c) = make_float4(dv[c], dv[c + 1], dv[c + 2], dv[c + 3]);
    }
    __syncthreads();
  }
  // relative-position-bias table: entry t = (dy + 7) * 15 + (dx + 7) collects dS[i][j] over the pairs with that offset
  float* tb = a.dtable_slab + (long)win * (RP * RP * a.heads);
  for (int t = i; t < RP * RP; t += 64) {
    const int dy = t / RP - (WS - 1), dx = t % RP - (WS - 1);
    float acc = 0.f;
    for (int yi = (dy > 0 ? dy : 0); yi < (dy < 0 ? WS + dy : WS); ++yi)
      for (int xi = (dx > 0 ? dx : 0); xi < (dx < 0 ? WS + dx : WS); ++xi)
        acc += Dm[yi * WS + xi][(yi - dy) * WS + (xi - dx)];
    tb[t * a.heads + h] = acc;
  }
}

// d(self.w): (w1, w2) = softmax(w), dsum = sum over (window, head) partials already reduced into g[2]
__global__ void winattn_wgrad_kernel(const float* __restrict__ w, const float* __restrict__ g, float* __restrict__ dw) {
  if (threadIdx.x == 0) {
    float w1, w2;
    mix_weights(w, w1, w2);
    const float dot = w1 * g[0] + w2 * g[1];
    dw[0] = w1 * (g[0] - dot);
    dw[1] = w2 * (g[1] - dot);
  }
}

int check(const UirWinAttnArgs& a, bool bwd) {
  UIR_REQUIRE(a.Q && a.K && a.V && a.table && a.w && a.B > 0 && a.heads > 0 && a.C > 0 && a.C % a.heads == 0);
  UIR_REQUIRE(a.ws == WS && a.H % WS == 0 && a.W % WS == 0 && a.shift >= 0 && a.shift < WS);
  const int hd = a.C / a.heads;
  UIR_REQUIRE(hd == 8 || hd == 16 || hd == 32 || hd == 64 || hd == 128);
  UIR_REQUIRE(a.ldq >= a.C && a.ldk >= a.C && a.ldv >= a.C && ((a.ldq | a.ldk | a.ldv) & 3) == 0);
  UIR_REQUIRE(uir_aligned16(a.Q) && uir_aligned16(a.K) && uir_aligned16(a.V));
  if (!bwd) UIR_REQUIRE(a.O && a.ldo >= a.C && (a.ldo & 3) == 0 && uir_aligned16(a.O));
  else {
    UIR_REQUIRE(a.dO && a.dQ && a.dK && a.dV && a.dtable_slab && a.dw_slab && a.ldo >= a.C && (a.ldo & 3) == 0);
    UIR_REQUIRE(a.lddq >= a.C && a.lddk >= a.C && a.lddv >= a.C && ((a.lddq | a.lddk | a.lddv) & 3) == 0);
    UIR_REQUIRE(uir_aligned16(a.dO) && uir_aligned16(a.dQ) && uir_aligned16(a.dK) && uir_aligned16(a.dV));
  }
  return UIR_OK;
}

}  // namespace

extern "C" int uir_winattn_fwd(const UirWinAttnArgs* p, void* stream) {
  const UirWinAttnArgs& a = *p;
  if (int rc = check(a, false)) return rc;
  const dim3 grid(a.B * (a.H / WS) * (a.W / WS), a.heads);
  hipStream_t st = static_cast<hipStream_t>(stream);
  switch (a.C / a.heads) {
    case 8: hipLaunchKernelGGL(winattn_fwd_kernel<8>, grid, dim3(64), 0, st, a); break;
    case 16: hipLaunchKernelGGL(winattn_fwd_kernel<16>, grid, dim3(64), 0, st, a); break;
    case 32: hipLaunchKernelGGL(winattn_fwd_kernel<32>, grid, dim3(64), 0, st, a); break;
    case 64: hipLaunchKernelGGL(winattn_fwd_kernel<64>, grid, dim3(64), 0, st, a); break;
    default: hipLaunchKernelGGL(winattn_fwd_kernel<128>, grid, dim3(64), 0, st, a); break;
  }
  return uir_launch_status();
}

extern "C" int uir_winattn_bwd(const UirWinAttnArgs* p, void* stream) {
  const UirWinAttnArgs& a = *p;
  if (int rc = check(a, true)) return rc;
  const dim3 grid(a.B * (a.H / WS) * (a.W / WS), a.heads);
  hipStream_t st = static_cast<hipStream_t>(stream);
  switch (a.C / a.heads) {            // 16 KB of chunk tiles + 33 KB for P and dS: under the 64 KB static limit for every head dim
    case 8: hipLaunchKernelGGL(winattn_bwd_kernel<8>, grid, dim3(64), 0, st, a); break;
    case 16: hipLaunchKernelGGL(winattn_bwd_kernel<16>, grid, dim3(64), 0, st, a); break;
    case 32: hipLaunchKernelGGL(winattn_bwd_kernel<32>, grid, dim3(64), 0, st, a); break;
    case 64: hipLaunchKernelGGL(winattn_bwd_kernel<64>, grid, dim3(64), 0, st, a); break;
    default: hipLaunchKernelGGL(winattn_bwd_kernel<128>, grid, dim3(64), 0, st, a); break;
  }
  return uir_launch_status();
}

extern "C" int uir_winattn_wgrad(const float* w, const float* dsum, float* dw, void* stream) {
  UIR_REQUIRE(w && dsum && dw);
  hipLaunchKernelGGL(winattn_wgrad_kernel, dim3(1), dim3(64), 0, static_cast<hipStream_t>(stream), w, dsum, dw);
  return uir_launch_status();
}
