// LDS-tiled f32 MFMA contraction for the WIDE layers of the NewBigFRFNModel / AST families: Linear
// layers with 256 ... 2048 input channels (src/model/block.py:223-282, src/Models/AST.py:294-326: LeFF / FRFN linear1 / linear2;
// AST.py:42-64 to_q / to_kv / proj) and their weight gradients.  uir_gemm keeps a whole weight panel of one column group in LDS
// (K <= ~768) and otherwise fell back to an un-staged kernel at 22-29 TFLOP/s; uir_gram's 64 x 64 output groups re-read their
// operands N / 64 resp. K / 64 times.  Here:
//
//     D[i][j] (+ epilogue) = sum_k A(i, k) * B(j, k)          128 x 128 output tile per workgroup, k in chunks of 32
//
// with either operand "k-contiguous" (activation rows [M][K], weights [N][K]) or "index-contiguous" (the token contraction of a
// weight gradient: A(n, m) = dY[m][n]; transposed weights W[k][n] of an input gradient).  Both operand chunks are staged through
// LDS with 16-byte global loads in the operand's own contiguous direction (next chunk in flight during the MFMAs, two LDS stages,
// one barrier per chunk); 4 waves as 2 x 2, each 64 x 64 = four v_mfma_f32_32x32x2_f32 accumulators: 64 MFMAs (4096 cycles) per
// 16 LDS reads per wave and chunk.  Split-k over blockIdx.y writes partial slabs (deterministic; summed by uir_reduce_slab).
#include "uir_common.h"
#include "../../include/uir_kernels.h"

namespace {

constexpr int TM = 128, TN = 128, TK = 32;
constexpr int KP = TK + 4;        // k-contiguous layout [128][36]: 16-byte rows, stride/4 odd -> conflict-free ds_read_b128
constexpr int IP = TM + 4;        // index-contiguous layout [32][132]
constexpr int STAGE = TM * KP;    // floats per operand per stage (>= 32 * IP)

// global -> registers: the operand's 128 x 32 chunk as 4 float4 per thread
template <bool KC>
struct Loader {
  const float* p[4];
  int kk[4];                                        // k offset of each float4 inside the chunk (KC: its first k; IC: its k)
  __device__ __forceinline__ void init(const float* base, long s_idx, long s_k, int idx0, int n_idx, int tid) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int e = tid + 256 * q;
      if (KC) {
        int r = idx0 + (e >> 3);
        r = r < n_idx ? r : n_idx - 1;              // clamped rows are computed and never stored
        kk[q] = 4 * (e & 7);
        p[q] = base + (long)r * s_idx;            // + k at load time
      } else {
        int i = idx0 + 4 * (e & 31);
        i = i + 4 <= n_idx ? i : n_idx - 4;
        kk[q] = e >> 5;
        p[q] = base + i;
      }
    }
  }
  __device__ __forceinline__ void load(float4 (&v)[4], int k0, int ke, long s_k) const {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int k = k0 + kk[q];
      const int kc = k < ke ? k : ke - 4;           // address stays inside the operand; the value is zeroed below
      const float4 t = *reinterpret_cast<const float4*>(KC ? p[q] + kc : p[q] + (long)kc * s_k);
      v[q] = k < ke ? t : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
  __device__ __forceinline__ void store(float* s, const float4 (&v)[4], int tid) const {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int e = tid + 256 * q;
      float* d = KC ? s + (e >> 3) * KP + 4 * (e & 7) : s + (e >> 5) * IP + 4 * (e & 31);
      *reinterpret_cast<float4*>(d) = v[q];
    }
  }
};

// one operand fragment of a 32-row tile for k sub-block jj: the four k values 8 jj + 4 h2 + e of this lane's row
template <bool KC>
__device__ __forceinline__ float4 frag(const float* s, int row, int jj, int h2) {
  if (KC) return *reinterpret_cast<const float4*>(s + row * KP + 8 * jj + 4 * h2);
  const float* c = s + (8 * jj + 4 * h2) * IP + row;
  return make_float4(c[0], c[IP], c[2 * IP], c[3 * IP]);
}

template <bool AKC, bool BKC>
__global__ __launch_bounds__(256, 2) void uir_tiled_kernel(const UirTiledArgs t) {
  extern __shared__ float smem[];
  float* As = smem;                       // [2][STAGE]
  float* Bs = smem + 2 * STAGE;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, h2 = lane >> 5;
  const int wr = wave >> 1, wc = wave & 1;
  const int tiles_j = (t.NN + TN - 1) / TN;
  const int i0 = (blockIdx.x / tiles_j) * TM, j0 = (blockIdx.x % tiles_j) * TN;
  const int split = blockIdx.y;
  const int ks = split * t.k_per_split;
  const int ke = ks + t.k_per_split < t.KK ? ks + t.k_per_split : t.KK;

  Loader<AKC> la;
  Loader<BKC> lb;
  la.init(t.A, t.sa_i, t.sa_k, i0, t.MM, tid);
  lb.init(t.B, t.sb_j, t.sb_k, j0, t.NN, tid);

  f32x16 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  float4 va[4], vb[4];
  la.load(va, ks, ke, t.sa_k);
  lb.load(vb, ks, ke, t.sb_k);
  la.store(As, va, tid);
  lb.store(Bs, vb, tid);
  __syncthreads();
  int stage = 0;
  for (int k0 = ks; k0 < ke; k0 += TK) {
    const bool more = k0 + TK < ke;
    if (more) {
      la.load(va, k0 + TK, ke, t.sa_k);
      lb.load(vb, k0 + TK, ke, t.sb_k);
    }
    const float* as = As + stage * STAGE;
    const float* bs = Bs + stage * STAGE;
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
      const float4 a0 = frag<AKC>(as, 64 * wr + li, jj, h2), a1 = frag<AKC>(as, 64 * wr + 32 + li, jj, h2);
      const float4 b0 = frag<BKC>(bs, 64 * wc + li, jj, h2), b1 = frag<BKC>(bs, 64 * wc + 32 + li, jj, h2);
      const float av[2][4] = {{a0.x, a0.y, a0.z, a0.w}, {a1.x, a1.y, a1.z, a1.w}};
      const float bv[2][4] = {{b0.x, b0.y, b0.z, b0.w}, {b1.x, b1.y, b1.z, b1.w}};
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
          for (int b = 0; b < 2; ++b) acc[a][b] = uir_mfma(av[a][e], bv[b][e], acc[a][b]);
    }
    if (more) {
      la.store(As + (stage ^ 1) * STAGE, va, tid);
      lb.store(Bs + (stage ^ 1) * STAGE, vb, tid);
    }
    __syncthreads();
    stage ^= 1;
  }

  float* D = t.D + (long)split * t.d_split;
  const bool gelu = t.act == UIR_ACT_GELU;
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const int col = j0 + 64 * wc + 32 * b + li;
      const float bj = (t.bias && col < t.NN) ? t.bias[col] : 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = i0 + 64 * wr + 32 * a + uir_mfma_row(r, h2);
        if (row < t.MM && col < t.NN) {
          float v = acc[a][b][r] + bj;
          if (t.pre_out) t.pre_out[(long)row * t.ldpre + col] = v;
          if (gelu) v = uir_gelu(v);
          if (t.R) v += t.R[(long)row * t.ldr + col];
          D[(long)row * t.sd_i + (long)col * t.sd_j] = v;
        }
      }
    }
}

}  // namespace

int uir_tiled_gemm(const UirTiledArgs& t, hipStream_t st) {
  UIR_REQUIRE(t.A && t.B && t.D && t.MM > 0 && t.NN > 0 && t.KK > 0 && t.k_per_split > 0 && (t.k_per_split & 3) == 0 && (t.KK & 3) == 0);
  const bool akc = t.sa_k == 1, bkc = t.sb_k == 1;
  UIR_REQUIRE(akc || t.sa_i == 1);
  UIR_REQUIRE(bkc || t.sb_j == 1);
  UIR_REQUIRE(uir_aligned16(t.A) && uir_aligned16(t.B));
  if (akc) UIR_REQUIRE((t.sa_i & 3) == 0); else UIR_REQUIRE((t.sa_k & 3) == 0 && (t.MM & 3) == 0 && t.MM >= 4);
  if (bkc) UIR_REQUIRE((t.sb_j & 3) == 0); else UIR_REQUIRE((t.sb_k & 3) == 0 && (t.NN & 3) == 0 && t.NN >= 4);
  UIR_REQUIRE(t.KK >= 4);
  const int nsplit = (t.KK + t.k_per_split - 1) / t.k_per_split;
  const dim3 grid(((t.MM + TM - 1) / TM) * ((t.NN + TN - 1) / TN), nsplit);
  const size_t lds = (size_t)4 * STAGE * sizeof(float);
  static_assert(STAGE >= TK * IP, "stage size");
#define UIR_TL(AK, BK) hipLaunchKernelGGL((uir_tiled_kernel<AK, BK>), grid, dim3(256), lds, st, t)
  if (akc && bkc) UIR_TL(true, true);
  else if (akc) UIR_TL(true, false);
  else if (bkc) UIR_TL(false, true);
  else UIR_TL(false, false);
#undef UIR_TL
  return uir_launch_status();
}
