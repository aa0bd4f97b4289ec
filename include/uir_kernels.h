/* uir_kernels.h - C ABI of libuir_hip.so: the MI355X (gfx950) kernels behind the
 * SpectralTransformer train step.
 *
 * The reference (KarthikSundar2002/Underwater-Image-Restoration) has no native layer: its
 * hot path is ATen ops called from src/Models/SpectralTransformer.py, src/Losses/losses.py
 * and src/ModelTrainer.py:78-90.  Each entry point below names the reference op site(s)
 * (path:line) whose arithmetic it replaces.  Conventions:
 *   - every pointer is a DEVICE pointer to fp32 data owned by the caller (PyTorch caching
 *     allocator tensors); kernels never allocate or free; scratch ("slab") is passed in;
 *   - activations are HWC rows: a [B,H,W,C] tensor is M = B*H*W rows of C floats with a row
 *     stride `ld*` (floats); rows must be 16-byte aligned where a field says "float4";
 *   - `stream` is a hipStream_t (NULL = default stream); every call is asynchronous;
 *   - return value: 0 = launched, UIR_EINVAL (1) = argument check failed (nothing launched),
 *     UIR_ELAUNCH (2) = HIP reported a launch error.  No exceptions cross this boundary.
 */
#ifndef UIR_KERNELS_H
#define UIR_KERNELS_H

#ifdef __cplusplus
extern "C" {
#endif

/* ---------------------------------------------------------------- uir_gemm */
enum { UIR_PRO_NONE = 0, UIR_PRO_LN = 1, UIR_PRO_GATE = 2 };
#define UIR_GEMM_SLAB_ROWS 512   /* max workgroups (= slab rows) of the GEMM variants with fused reductions */
enum { UIR_EPI_PLAIN = 0, UIR_EPI_GATE_BWD = 1, UIR_EPI_LEAKY_BWD = 2, UIR_EPI_GATE_BWD_DW = 3, UIR_EPI_GELU_BWD = 4, UIR_EPI_PLAIN_DW = 5, UIR_EPI_LN_BWD = 6 };
/* UIR_EPI_LN_BWD (round 5): Y = R + LayerNorm'(X . W^T) - the backward of the channel LayerNorm in front of a 1x1 convolution as the epilogue of the
 * contraction that produces its output gradient (dt1 / dt2 of a TransformerBlock at C = 64 / 128, SpectralTransformer.py:144-147): G = the LayerNorm's
 * input (ldg), stats = its saved (mean, rstd) per row, ln_g = its weight, R = the residual path's gradient (optional); dw_slab rows [row tile][2 N] =
 * dgamma | dbeta partials (at most M / 64 rows are written: zero-fill the rest once), dw_stride >= 2 N.  N = 64 or 128 (= the whole row), 4096 <= M <=
 * 65536, K >= 64, no bias / activation / column groups; the arithmetic of uir_layernorm_bwd.  Any other shape is UIR_EINVAL. */
/* UIR_EPI_PLAIN_DW (round 4): Y = X . W^T as PLAIN (no bias / activation / residual) and, from the same pass over X, the per-batch Gram
 *   dW[z][k][j] = sum_m X[z][m][k] * G[z][m][j]   (K <= 32 channels of X, j < dw_cols <= 32 columns of a second tensor G [batches][M][ldg])
 * as per-workgroup partials dw_slab[blockIdx.x][z * K * dw_cols + k * dw_cols + j] (summed by uir_reduce_slab).  MDTA's value path: dvf = dx1 . WB
 * with dW_B = dx1^T vf, and dv = du . WA with dW_A = du^T v (SpectralTransformer.py:100,113 backward) - X is read once instead of twice. */
enum { UIR_ACT_NONE = 0, UIR_ACT_LEAKY = 1, UIR_ACT_GELU = 2 };   /* GELU: exact-erf form; plain epilogue only */

/* Dense channel contraction over HWC rows, 1 or 9 (3x3, pad 1) taps:
 *   Y[z][m][n] = epi( sum_tap sum_k pro(X[z][nbr(m,tap)][k]) * W[z][n][k][tap] + bias[n] ) + R[z][m][n]
 * Replaces: nn.Conv2d 1x1 sites SpectralTransformer.py:81,83,86,90,122,125,165-172,206,234-235;
 * 3x3 dense sites :154,194,219,247,252; their input-gradients (swap w_sn/w_sk, flip=1);
 * LayerNorm fused as a prologue (:137,139,144-147); GELU gate fused as a prologue (:128-129). */
typedef struct UirGemmArgs {
  const float* X;      /* [batches][M][ldx], float4-aligned rows                                */
  const float* W;      /* weights, element (n,k,tap) at W[z*w_bs + n*w_sn + k*w_sk + tap*w_st]  */
  float* Y;            /* [batches][M][ldy]                                                      */
  const float* bias;   /* [real N] or NULL                                                       */
  const float* R;      /* residual added after epi (PLAIN only), may alias Y; NULL = none        */
  const float* ln_g;   /* PRO_LN: gamma[K], beta[K] (16-B aligned)                               */
  const float* ln_b;
  float* stats;        /* PRO_LN: optional out [batches*M][2] = (mean, rstd) for backward        */
  const float* G;      /* GATE_BWD: forward gate input g (g1 at col n, g2 at col g_off+n);       */
                       /* LEAKY_BWD: forward LeakyReLU output                                    */
  long x_bs, y_bs, r_bs, g_bs, w_bs; /* per-batch strides (floats); w_bs=0: shared weights      */
  long w_sn, w_sk, w_st;
  int ldx, ldy, ldr, ldg;
  int M, N, K;         /* rows per batch; padded output cols; padded contraction length          */
  int batches;         /* grid z                                                                 */
  int taps, flip;      /* taps 1|9; flip=1 uses weight tap 8-t (input-gradient of a 3x3 conv)    */
  int H, Wd;           /* spatial dims (taps==9); M % (H*Wd) == 0                                */
  int n_grp_real, n_grp_pad; /* output col n -> real weight row: groups of n_grp_pad cols, the   */
  int k_grp_real, k_grp_pad; /* first n_grp_real real, the rest exact-zero; pad<=0: identity     */
  int pro, gate_off;   /* prologue; GATE: second operand at column gate_off + k                  */
  int epi, g_off;
  int act;             /* applied before epi                                                     */
  float slope;         /* LeakyReLU negative slope                                               */
  float ln_eps;
  float* dw_slab;      /* GATE_BWD_DW / PLAIN_DW: per-workgroup partials [gridDim.x][dw_stride] of the project_out     */
  long dw_stride;      /* weight gradient dW[k][n] = sum_m X[m][k] * gate(G)[m][n], summed by uir_reduce_slab;*/
                       /* gridDim.x = min(M/128, UIR_GEMM_SLAB_ROWS).  Requires K <= 32, N <= 96, M % 128 == 0 */
                       /* GELU_BWD: Y = (X.W^T) * gelu'(G + bias) (bias = the bias INSIDE the GELU, not added to the  */
                       /* product) and per-workgroup column sums of Y to dw_slab rows: LeFF's Linear2 input gradient   */
                       /* with the second GELU's backward and the depthwise bias gradient (src/model/block.py:240-246) */
  int tap0, ntap;      /* taps == 9 only: contract taps [tap0, tap0 + ntap) (ntap <= 0: all nine).  uir_gemm  */
                       /* itself uses this to run a 3x3 convolution whose weight panel exceeds LDS as a few   */
                       /* launches over tap groups that accumulate into Y (R = Y)                             */
  float* pre_out;      /* UIR_ACT_GELU only, optional: the pre-activation X.W^T + bias is ALSO stored here (row stride */
  long ldpre;          /* ldpre) - what the GELU backward needs; one pass instead of GEMM + element-wise (FRFN / LeFF  */
                       /* linear1, block.py:252,273).  batches == 1                                                   */
  const float* out_scale; /* optional (with R; PLAIN epilogue, taps 1, batches 1): Y = R + out_scale[row / scale_rows] * act(X.W^T + bias) -  */
  int scale_rows;      /* (>= 32) the per-sample DropPath factor of a residual branch (timm drop_path: src/Models/AST.py:617,620,        */
                       /* src/model/model.py:91) folded into the branch's last Linear; scale_rows = token rows per sample        */
  int dw_cols;         /* EPI_PLAIN_DW: columns of G that enter the fused Gram (<= 32)                                                   */
  int dw_rows;         /* EPI_PLAIN_DW: workgroups per batch = slab rows written (1 .. M / 128); the caller reduces exactly these        */
  int il16;            /* EPI_GATE_BWD / EPI_GATE_BWD_DW only: 1 = G and Y hold the GDFN tensors in the chunk-interleaved layout of      */
                       /* UirGdfnFwdArgs.il (hidden channel n at column 32 (n / 16) + n % 16, its x2 partner g_off = 16 columns on)      */
} UirGemmArgs;
int uir_gemm(const UirGemmArgs* args, void* stream);

/* ---------------------------------------------------------------- uir_gram */
enum { UIR_GRAM_PRO_NONE = 0, UIR_GRAM_PRO_LN = 1, UIR_GRAM_PRO_GATE = 2 };

/* Contraction over pixels (weight gradients and the MDTA q.k^T Gram matrix):
 *   out[z][n][k][tap] = sum_m A[z][m][n] * pro(B[z][nbr(m,tap)][k])
 * computed as per-chunk partial slabs slab[chunk][...] (deterministic; summed by uir_reduce_slab).
 * Optional extras per chunk: column sums of squares of A and B (the L2 norms of F.normalize,
 * SpectralTransformer.py:98) and column sums of A (bias gradients).
 * Replaces: torch.matmul(q, k^T) (:99) and every conv weight-gradient of autograd's
 * convolution_backward on the path. */
typedef struct UirGramArgs {
  const float* A;      /* [batches][M][lda]                                                      */
  const float* B;      /* [batches][M][ldb]                                                      */
  float* slab;         /* [nchunks][slab_stride]                                                 */
  const float* stats;  /* PRO_LN: [batches*M][2] (mean, rstd) written by uir_gemm                */
  const float* ln_g;
  const float* ln_b;
  long a_bs, b_bs;     /* per-batch strides (floats)                                             */
  long o_sn, o_sk, o_st, o_bs; /* output strides: real n, real k, tap, batch                     */
  long slab_stride;    /* floats per chunk                                                       */
  long sq_a_off, sq_b_off, cs_a_off; /* offsets (floats) inside a chunk slab; <0 = not computed  */
  int lda, ldb;
  int M, N, K;         /* rows per batch; padded cols of A; padded cols of B                     */
  int batches;
  int n_grp_real, n_grp_pad, k_grp_real, k_grp_pad;
  int taps, H, Wd;
  int pro_b, gate_off;
  int rows_per_chunk;  /* multiple of 8                                                          */
  int nchunks;
} UirGramArgs;
int uir_gram(const UirGramArgs* args, void* stream);

/* out[i] = (accumulate ? out[i] : 0) + scale * sum_c slab[c*stride + i], i < n */
int uir_reduce_slab(const float* slab, int nchunks, long stride, float* out, long n, int accumulate,
                    float scale, void* stream);
/* The same reduction for MANY tensors in one launch: descs[d] (device memory) describes one uir_reduce_slab call,
 * first_block[d] (device, ascending, first_block[0] = 0) is the first workgroup of descriptor d (64 outputs per workgroup),
 * total_blocks their sum.  Summation order per output is exactly uir_reduce_slab's: results are bit-identical.  The train
 * step batches the weight-gradient reductions of one network stage this way (~380 launches per step -> ~75). */
typedef struct UirReduceDesc {
  const float* slab;
  float* out;
  long stride;
  long n;
  int nchunks;
  int accumulate;
  float scale;
  int _pad;
} UirReduceDesc;
int uir_reduce_slab_batch(const UirReduceDesc* descs_dev, const int* first_block_dev, int ndesc, int total_blocks, void* stream);

/* ------------------------------------------------------- uir_conv3x3_split */
/* Dense 3x3 convolution (pad 1, no bias) with >= 32 channels on both sides as an implicit GEMM on split-bf16 MFMA
 * (csrc/conv3b.hip): UpSample1's Conv2d(C, 2C, 3) forward, its input gradient and its weight gradient
 * (SpectralTransformer.py:191-198; autograd's convolution_backward for :196).  f32-accurate: every f32 operand is the exact
 * sum of three bf16 planes and six of the nine plane products are accumulated in f32 (csrc/uir_common.h).
 *
 * uir_conv3x3_wsplit  native weight W[Cout][Cin][3][3] -> bf16 planes, once per optimizer step (Cout % 16 == 0, Cin % 16 == 0).  A layer with
 *   N outputs and K inputs is stored as planes[pl][tap][K / 16][N][16] - the 32 x 16 weight fragment of one MFMA step is 1 KB contiguous:
 *     planes_fwd (N = Cout, K = Cin):   element (tap,     n = cout, k = cin ) = plane pl of W[cout][cin][tap]
 *     planes_bwd (N = Cin,  K = Cout):  element (8 - tap, n = cin,  k = cout) = plane pl of W[cout][cin][tap]
 *   each 3 * 9 * Cout * Cin 16-bit values; either pointer may be NULL.
 * uir_conv3x3_split   Y[p][n] = (R ? R[p][n] : 0) + sum_tap sum_k X[nbr(p, tap)][k] * Wp(tap, n, k)  over HWC pixel rows
 *   p = (b * H + y) * Wd + x (zero outside the image).  H % 4 == 0, Wd % 16 == 0, N % 32 == 0, K % 32 == 0; rows 16-byte aligned.
 * uir_conv3x3_split_wgrad  slab[c][(n * K + k) * 9 + tap] = sum over the pixels p of chunk c of dY[p][n] * X[nbr(p, tap)][k]
 *   (the native weight layout per slab row; uir_reduce_slab sums the rows).  N % 64 == 0, K % 32 == 0, H % 4 == 0,
 *   Wd % 16 == 0.  `nchunks` (<= B * H/4 * Wd/16) slab rows are written, one per workgroup column. */
typedef struct UirConv3bArgs {
  const float* X;            /* [B*H*Wd][ldx]                                                      */
  const unsigned short* Wp;  /* bf16 planes [3][9][K / 16][N][16] (uir_conv3x3_wsplit)            */
  float* Y;                  /* [B*H*Wd][ldy]                                                      */
  const float* R;            /* optional residual [B*H*Wd][ldr]; may alias Y                       */
  int ldx, ldy, ldr;
  int B, H, Wd;
  int N, K;
  int tile;                  /* 0 = chosen by the library; else 1000 * pixels + columns of the workgroup tile (tests, tuning): */
                             /* 256032 (H % 16), 128032, 128064, 128128 (H % 8), 64064, 64128                                  */
} UirConv3bArgs;
typedef struct UirConv3bWgradArgs {
  const float* dY;           /* [B*H*Wd][ldy], N columns used                                      */
  const float* X;            /* [B*H*Wd][ldx], K columns used                                      */
  float* slab;               /* [rows][slab_stride], slab_stride >= N * K * 9                      */
  long slab_stride;
  int ldy, ldx;
  int B, H, Wd;
  int N, K;
  int nchunks;
} UirConv3bWgradArgs;
int uir_conv3x3_wsplit(const float* W, int Cout, int Cin, unsigned short* planes_fwd, unsigned short* planes_bwd, void* stream);
int uir_conv3x3_split(const UirConv3bArgs* args, void* stream);
int uir_conv3x3_split_wgrad(const UirConv3bWgradArgs* args, void* stream);

/* -------------------------------------------------------------- uir_dwconv */
/* Depthwise 3x3, pad 1, no bias, on HWC rows; C (padded) % 4 == 0.  flip=1 gives the
 * input-gradient.  Replaces nn.Conv2d(groups=C) sites SpectralTransformer.py:82,89,123-124. */
typedef struct UirDwArgs {
  const float* X;      /* [B*H*W][ldx] float4 rows                                               */
  const float* Wt;     /* native Conv2d layout [Creal][1][3][3]                                  */
  float* Y;            /* [B*H*W][ldy]                                                           */
  int ldx, ldy;
  int B, H, Wd, C;     /* C = padded channel count                                               */
  int grp_real, grp_pad; /* padded channel -> real weight row (exact-zero pads)                  */
  int flip;
  /* LeFF / FRFN (src/model/block.py:234-247, src/Models/AST.py:294-326): Linear -> GELU -> depthwise 3x3 (+ bias) -> GELU.
   * act_in = 1: X holds the PRE-activation of the first GELU, which is applied on load (the activated tensor is never stored);
   * G != NULL: second output G = gelu(Y + bias[c]) written in the same pass (Y, the pre-activation, is kept for the backward). */
  int act_in;
  const float* bias;   /* [C], with G                                                            */
  float* G;            /* [B*H*W][ldg] or NULL                                                   */
  int ldg;
} UirDwArgs;
int uir_dwconv(const UirDwArgs* args, void* stream);

/* weight-gradient partials: slab[block][Creal][9] = sum_pixels dY[p][c] * X[nbr(p,t)][c] */
typedef struct UirDwWgradArgs {
  const float* dY; const float* X; float* slab;
  int ldy, ldx;
  int B, H, Wd, C;
  int grp_real, grp_pad;
  int creal;           /* number of real channels (slab row count)                               */
  int nblocks;         /* grid size = slab chunk count                                           */
} UirDwWgradArgs;
int uir_dwconv_wgrad(const UirDwWgradArgs* args, void* stream);

/* fused backward of the depthwise conv: dX = conv(dY, W flipped) AND slab[block][Creal][9] partials of dW
 * in one pass (dY read once).  Autograd's convolution_backward of the three depthwise sites above. */
typedef struct UirDwBwdArgs {
  const float* dY; const float* X; const float* Wt; float* dX; float* slab;
  int ldy, ldx, ldo;
  int B, H, Wd, C;
  int grp_real, grp_pad;
  int creal;
  int nblocks;
  /* act = 1: X holds the pre-activation p of a GELU in front of the convolution (LeFF / FRFN): the weight gradient uses gelu(p)
   * and dX is the gradient with respect to p, i.e. conv^T(dY) * gelu'(p) - Linear1's GELU backward in the same pass; db_slab
   * (optional, [nblocks][creal]) then receives the partial column sums of dX = the gradient of Linear1's bias. */
  int act;
  float* db_slab;
  /* U != NULL (with act = 1; LeFF, src/model/block.py:240-244): dY is the gradient of G = gelu(U + gbias), the second GELU behind the
   * convolution: dY is multiplied by gelu'(U + gbias) as it is loaded, so the gradient of the convolution output is never stored;
   * dbu_slab (optional, [nblocks][creal]) receives its partial column sums = the gradient of the depthwise bias. */
  const float* U;
  const float* gbias;
  float* dbu_slab;
  int ldu;
} UirDwBwdArgs;
int uir_dwconv_bwd_fused(const UirDwBwdArgs* args, void* stream);

/* ---------------------------------------------------------------- uir_gdfn */
/* Tile-fused GDFN forward of one TransformerBlock:
 *   Y = X + W_out . ( gelu(dw3x3(W_in . LN(X))[:h]) * dw3x3(W_in . LN(X))[h:] )
 * One pass over the pixels instead of project_in -> conv -> gate/project_out as three kernels.
 * Replaces SpectralTransformer.py:127-130 (GDFN.forward) together with norm2 and the residual
 * add of :147-148.  T1 / T2: the pre- / post-depthwise tensors the backward kernels read (both, T2 alone, or neither): x1 half at
 * column c, x2 half at column hp + c, pad columns c in [h, hp) zero.  T2 alone (round 5, what the C <= 32 train step passes): the
 * pre-depthwise tensor is not kept - uir_dwln_bwd (XP = NULL) recomputes the values it needs. */
typedef struct UirGdfnFwdArgs {
  const float* X;      /* [B*H*W][ldx] block input (also the residual), float4-aligned rows     */
  const float* ln_g;   /* norm2 gamma[C], beta[C] (16-B aligned)                                 */
  const float* ln_b;
  const float* Win;    /* project_in  [2h][C]   (16-B aligned)                                   */
  const float* Wdw;    /* conv        [2h][1][3][3]                                              */
  const float* Wout;   /* project_out [C][h]                                                     */
  float* Y;            /* [B*H*W][ldy]                                                           */
  float* T1;           /* optional [B*H*W][ldt] (only together with T2)                          */
  float* T2;           /* optional [B*H*W][ldt]                                                  */
  float* stats;        /* optional [B*H*W][2] = (mean, rstd) of norm2 for the backward           */
  int ldx, ldy, ldt;
  int B, H, Wd;
  int C;               /* in {16,32,64,128}                                                      */
  int h, hp;           /* hidden = int(2.66 C); hp = h padded to a multiple of 16                */
  float ln_eps;
  int il;              /* column layout of T1 / T2.  0: x1 half at column j, x2 half at hp + j.           */
                       /* 1: interleaved by 16-channel chunk - hidden channel j's x1 value at column      */
                       /* 32 (j / 16) + j % 16, its x2 value 16 columns on: every (pixel, chunk) access   */
                       /* of this kernel, the gate-backward GEMM and uir_dwln_bwd is ONE 128-byte line    */
                       /* instead of two 64-byte halves (3.5 vs 5-6.5 TB/s: tools/probes/vmem/halfline)   */
} UirGdfnFwdArgs;
int uir_gdfn_fwd(const UirGdfnFwdArgs* args, void* stream);

/* Tile-fused front of an MDTA block: Y = dw3x3(W_qkv . LN(X)) (SpectralTransformer.py:93-94 with norm1 of :144),
 * Ypre = W_qkv . LN(X) (saved for the depthwise weight gradient).  3C output channels, chunks of 32. */
typedef struct UirQkvFwdArgs {
  const float* X;      /* [B*H*W][ldx] block input, float4-aligned rows                           */
  const float* ln_g;   /* norm1 gamma[C], beta[C] (16-B aligned)                                  */
  const float* ln_b;
  const float* Wqkv;   /* qkv      [3C][C]   (16-B aligned)                                       */
  const float* Wdw;    /* qkv_conv [3C][1][3][3]                                                  */
  float* Ypre;         /* [B*H*W][ldp] pre-depthwise, or NULL: not stored (uir_dwln_bwd, XP = NULL, recomputes it) */
  float* Y;            /* [B*H*W][ldq] q | k | v                                                  */
  float* stats;        /* optional [B*H*W][2] = (mean, rstd) of norm1 for the backward            */
  int ldx, ldp, ldq;
  int B, H, Wd;
  int C;               /* in {16,32,64,128}                                                       */
  float ln_eps;
  /* optional (C <= 32), round 4: the per-sample Gram of MDTA's attention logits from the tile that is on chip anyway
   * (SpectralTransformer.py:96-98: q k^T and the norms F.normalize divides by).  Row `tile` of gram_slab (tile = (b * tiles_y + ty) *
   * tiles_x + tx, 12 x 16 pixel tiles) receives  G[C][C] = sum_px q[px][m] k[px][n]  |  sum_px q[px][m]^2 [C]  |  sum_px k[px][n]^2 [C];
   * the caller sums a sample's rows (uir_reduce_slab / _batch).  NULL: not computed. */
  float* gram_slab;
  long gram_stride;    /* floats between rows, >= C * C + 2 * C                                    */
} UirQkvFwdArgs;
int uir_qkv_fwd(const UirQkvFwdArgs* args, void* stream);

/* Tile-fused pixel-sized tail of MDTA: U = V . WA[b]^T, VF = dw3x3(U), Y = VF . WB[b]^T + R with the per-sample
 * C x C matrices of uir_attn_fwd (SpectralTransformer.py:100,107,110,113 after the folding of DESIGN.md section 5;
 * R = the block input: residual of :146). */
typedef struct UirAttnvFwdArgs {
  const float* V;      /* [B*H*W][ldv] v slice of qkv, float4-aligned rows                        */
  const float* WA;     /* [B][C][C]                                                               */
  const float* WB;     /* [B][C][C]                                                               */
  const float* Wdw;    /* kv_conv weights of the vf half [C][1][3][3]                             */
  const float* R;      /* [B*H*W][ldr] residual                                                   */
  float* U;            /* [B*H*W][ldu]                                                            */
  float* VF;           /* [B*H*W][ldf]                                                            */
  float* Y;            /* [B*H*W][ldy]                                                            */
  int ldv, ldr, ldu, ldf, ldy;
  int B, H, Wd;
  int C;               /* in {16,32,64,128}                                                       */
} UirAttnvFwdArgs;
int uir_attnv_fwd(const UirAttnvFwdArgs* args, void* stream);


/* uir_attnv_bwd (round 5): the backward of uir_attnv_fwd in one pass over the pixels, C = 16 / 32 (H % 8 == 0, Wd % 16 == 0) or C = 64 (H % 8 == 0, Wd % 8 == 0)
 * (SpectralTransformer.py:100,107,110,113 under autograd):
 *   dV[p] = dw3x3^T(dX1 . WB[b])[p] . WA[b]                                     written to dV rows (ldd)
 *   slab_ab[w][b*C*C + i*C + j]       = partial of dWB[b][i][j] = sum_p dX1[p][i] VF[p][j]
 *   slab_ab[w][nab + b*C*C + j*C + o] = partial of dWA[b][j][o] = sum_p du[p][j] V[p][o]      w < uir_attnv_bwd_rows_c(H, Wd, C) rows, stride slab_ab_stride >= 2 nab
 *   slab_dw[b * rows + w][c*9 + t]    = partial of the depthwise weight gradient sum_p dvf[p - off(t)][c] U[p][c]
 * with nab = B*C*C; uir_reduce_slab sums the rows.  Replaces the dvf GEMM (+ fused Gram), uir_dwconv_bwd_fused and the dv GEMM (+ fused Gram).
 * Split-bf16 form only (UIR_FUSED_SPLIT=1), otherwise UIR_EINVAL. */
typedef struct UirAttnvBwdArgs {
  const float* dX1;    /* [B*H*W][ldx] gradient at the value path's output                         */
  const float* WA;     /* [B][C][C]                                                               */
  const float* WB;     /* [B][C][C]                                                               */
  const float* Wdw;    /* kv_conv weights of the vf half [C][1][3][3]                             */
  const float* U;      /* [B*H*W][ldu] saved by uir_attnv_fwd                                     */
  const float* VF;     /* [B*H*W][ldf] saved by uir_attnv_fwd                                     */
  const float* V;      /* [B*H*W][ldv] v slice of qkv                                             */
  float* dV;           /* [B*H*W][ldd]                                                            */
  float* slab_ab;
  float* slab_dw;
  long slab_ab_stride, nab;
  int ldx, ldu, ldf, ldv, ldd;
  int B, H, Wd;
  int C;               /* 16, 32 or 64                                                            */
} UirAttnvBwdArgs;
int uir_attnv_bwd(const UirAttnvBwdArgs* args, void* stream);
int uir_attnv_bwd_rows(int H, int Wd);   /* host helper: slab rows per image at C <= 32 (not a status) */
int uir_attnv_bwd_rows_c(int H, int Wd, int C);   /* host helper: slab rows per image for any supported C (not a status) */

/* ------------------------------------------------------------- uir_dwln_bwd */
/* Backward of  LayerNorm -> 1x1 conv (C -> CH, no bias) -> depthwise 3x3 (no bias)  in one pass over the pixels (csrc/dwln.hip):
 *   dP = dw3x3^T(G) (kept on chip),  dX = dRes + LayerNorm'(dP . W1),  and the partial sums of all four parameter gradients.
 * Replaces autograd's backward of norm2 -> ffn.project_in -> ffn.conv (SpectralTransformer.py:122-124,147; hp > 0: the GDFN
 * column layout, x1 half at columns [0, h), x2 half at [hp, hp + h), weight rows [0, h) and [h, 2h)) and of
 * norm1 -> attn.qkv -> attn.qkv_conv (:81-82,94,144; hp = 0: ch consecutive channels).
 * slab[block][slab_stride] = dWdw (CHreal*9, Conv2d layout [CHreal][1][3][3]) | dW1 (CHreal*C, [CHreal][C]) | dgamma (C) | dbeta (C),
 * CHreal = 2h or ch; uir_reduce_slab sums the nblocks rows.  Every workgroup writes its whole row. */
typedef struct UirDwLnBwdArgs {
  const float* G;      /* [B*H*W][ldg] gradient of the depthwise output, float4-aligned rows                 */
  const float* XP;     /* [B*H*W][ldp] depthwise input saved by the forward (same column layout as G), or NULL */
                       /* (round 5): the kernel recomputes its centre values LN(X) . W1^T per 32-channel chunk  */
                       /* (one f32 MFMA tile per wave from the tile it normalises anyway); ldp is ignored then */
  const float* Wdw;    /* [CHreal][9]                                                                       */
  const float* W1;     /* [CHreal][C]                                                                       */
  const float* X;      /* [B*H*W][ldx] LayerNorm input                                                      */
  const float* stats;  /* [B*H*W][2] (mean, rstd) written by the forward                                    */
  const float* gamma;  /* [C]                                                                               */
  const float* beta;   /* [C]                                                                               */
  const float* dRes;   /* [B*H*W][ldr] gradient arriving on the residual path                               */
  float* dX;           /* [B*H*W][ldo]                                                                      */
  float* slab;
  long slab_stride;
  int ldg, ldp, ldx, ldr, ldo;
  int B, H, Wd;
  int C;               /* 16 or 32                                                                          */
  int h, hp;           /* GDFN layout (hp > 0, multiple of 16): (C, hp) in {(32, 96), (16, 48)}             */
  int ch;              /* consecutive layout (hp == 0): (C, ch) in {(32, 96), (16, 48)}                     */
  int nblocks;         /* persistent workgroups = slab rows                                                 */
  int il;              /* GDFN layout only: 1 = G / XP columns interleaved by chunk (UirGdfnFwdArgs.il)             */
} UirDwLnBwdArgs;
int uir_dwln_bwd(const UirDwLnBwdArgs* args, void* stream);

/* ----------------------------------------------------------- uir_layernorm */
/* Input-gradient of the channel LayerNorm (+ residual-path gradient) and per-block partials
 * of dgamma/dbeta: dX = dRes + LN'(dT).  slab[block][2*C] = (dgamma | dbeta).
 * Replaces layer_norm backward of SpectralTransformer.py:144-147. */
typedef struct UirLnBwdArgs {
  const float* dT; const float* X; const float* stats; const float* gamma;
  const float* dRes;   /* may be NULL                                                            */
  float* dX; float* slab;
  int ldt, ldx, ldr, ldo;
  int M, C;            /* C in {16,32,64,128,256,512}                                            */
  int nblocks;
} UirLnBwdArgs;
int uir_layernorm_bwd(const UirLnBwdArgs* args, void* stream);

/* ---------------------------------------------------------------- uir_attn */
/* Channel attention of MDTA (SpectralTransformer.py:98-100, 107, 113), algebraically folded:
 * per batch b the d x d maps attn[b,h] are composed with the 1x1 projections so that
 *   u   = WA[b] v      with WA[b] = P * blockdiag(attn[b]),   P = W_kv[vf half] * W_project_out
 *   out = WB[b] vf     with WB[b] = W_project_outf * blockdiag(attn[b])
 * (`attn @ v`, project_out, kv and `attn @ vf`, project_outf become two per-batch 1x1 GEMMs).
 * uir_attn_fwd: Gram matrix + squared norms -> attn (softmax(normalize(q) normalize(k)^T * T)),
 * WA, WB.  One workgroup per batch. */
typedef struct UirAttnFwdArgs {
  float* G;            /* [B][C][C]   sum_p q_i k_j         (input; OUTPUT - the head blocks - when `slab` is given) */
  float* sq_q;         /* [B][C]      sum_p q_i^2           (same)                                */
  float* sq_k;         /* [B][C]                            (same)                                */
  const float* temperature; /* [heads]                                                           */
  const float* P;      /* [C][C]                                                                 */
  const float* Wof;    /* [C][C] project_outf.weight                                             */
  float* attn;         /* [B][heads][d][d]                                                       */
  float* WA;           /* [B][C][C]                                                              */
  float* WB;           /* [B][C][C]                                                              */
  int B, C, heads;
  /* optional: sum the uir_gram partial rows here instead of a separate uir_reduce_slab launch (the kernel sits on the  */
  /* forward chain 24 times per step): row c of `slab` holds [G (B*C*C) ... sq_q at sq_q_off (B*C) ... sq_k at sq_k_off];  */
  /* the reduced head blocks of G and the norms are written to G / sq_q / sq_k for uir_attn_bwd.  NULL: G etc. are inputs */
  const float* slab; int nch; long slab_stride; long sq_q_off, sq_k_off;
} UirAttnFwdArgs;
int uir_attn_fwd(const UirAttnFwdArgs* args, void* stream);

/* Backward of the above.  Inputs dWA[b], dWB[b] (per-batch Grams of the two GEMMs); outputs
 * Mqk[b] ([2C][2C]: d[q;k] = Mqk[b] [q;k]), and per-batch partials dP[b], dWof[b], dT[b]. */
typedef struct UirAttnBwdArgs {
  const float* dWA; const float* dWB;       /* [B][C][C]                                         */
  const float* G; const float* sq_q; const float* sq_k; const float* temperature;
  const float* attn; const float* P; const float* Wof;
  float* Mqk;          /* [B][2C][2C]                                                            */
  float* dP;           /* [B][C][C]                                                              */
  float* dWof;         /* [B][C][C]                                                              */
  float* dT;           /* [B][heads]                                                             */
  int B, C, heads;
  /* optional: dWA / dWB as uir_gram partial rows (row c: dWA[b] at dWA_off + b*C*C, dWB[b] at dWB_off + b*C*C), summed   */
  /* here; `dWA` / `dWB` are then ignored                                                                                 */
  const float* slab; int nch; long slab_stride; long dWA_off, dWB_off;
} UirAttnBwdArgs;
int uir_attn_bwd(const UirAttnBwdArgs* args, void* stream);

/* ------------------------------------------------------------ uir_spectral */
/* One axis of a 2-D DFT of [B,H,W,C] tensors held as separate re/im planes, as an f32-MFMA
 * contraction with cos/sin twiddle tables (any length N, no radix restriction):
 *   out[o][k][i][c] = scale * sum_n e^{sign*2*pi*i*k*n/N} in[o][n][i][c]
 * element (o,n,i,c) at ((o*N + n)*I + i)*C + c.  in_im NULL = real input; out_im NULL = only the
 * real part is produced.  Replaces torch.fft.fft2 / ifft2 (SpectralTransformer.py:176, 186) and
 * their adjoints. */
typedef struct UirDftArgs {
  const float* in_re; const float* in_im;
  float* out_re; float* out_im;
  const float* cos_t;  /* [N][N] cos(2*pi*k*n/N) (symmetric)                                     */
  const float* sin_t;  /* [N][N] sin(2*pi*k*n/N)                                                 */
  int O, N, I, C;
  float sign;          /* -1 forward, +1 inverse                                                 */
  float scale;
} UirDftArgs;
int uir_dft_axis(const UirDftArgs* args, void* stream);

/* Same transform for N = 16*N2 in {32, 64, 128} as a two-stage (16 x N2) Cooley-Tukey factorisation on the
 * matrix cores (4x fewer MFMAs at N = 128).  ta / tb are host-built (fp64) MFMA operand tables for this
 * (N, sign): ta[16][64] = stage-A operand per (k-step, lane); tb[8][2*N2][64] = stage-B block-diagonal operand
 * per (k1 pair, k-step, lane) with the inter-stage twiddle folded in (layout: csrc/spectral.hip, engine.dft2_tables). */
/* Round 4: the element-wise neighbours of UpSample's transforms (SpectralTransformer.py:176-186 and their adjoints) ride on the
 * transform that produces / consumes their operand, so the complex planes between them never reach HBM:
 *   pro  UIR_DFT_PRO_POLAR       input z = in_re * exp(i in_im)                       (torch.complex(Mag cos, Mag sin), :183-185)
 *        UIR_DFT_PRO_CABS_BWD    input z = in_re / aux2 * (aux0 + i aux1), 0 where aux2 == 0     (adjoint of |z|: in_re = g_t,
 *                                aux0 / aux1 / aux2 = Re z, Im z, |z| saved by the forward)
 *        UIR_DFT_PRO_MAGPHA_BWD  input (g_re, g_im) of uir_magpha_bwd from in_re = g_mag, in_im = g_pha, aux0 = mag, aux1 = pha
 *   epi  UIR_DFT_EPI_MAGPHA      out_re = |F|, out_im = atan2(Im, Re), Im := +0 at the self-conjugate bins (uir_magpha_fwd; the
 *                                transform must be the H axis of a [B][H][W][C] tensor: O = B, N = H, I = W)
 *        UIR_DFT_EPI_CABS        out_re / out_im as usual and aux_out = |out|                                    (:186)
 *        UIR_DFT_EPI_POLAR_BWD   out_re = g_mag, out_im = g_pha of uir_polar_bwd with aux0 = mag, aux1 = pha at the output element
 *   pro = epi = 0 with out_im == NULL (real output): aux0, if given, is ADDED to the result (out_re = scale * Re X + aux0; aux0 may be out_re) */
#define UIR_DFT_PRO_NONE 0
#define UIR_DFT_PRO_POLAR 1
#define UIR_DFT_PRO_CABS_BWD 2
#define UIR_DFT_PRO_MAGPHA_BWD 3
#define UIR_DFT_EPI_NONE 0
#define UIR_DFT_EPI_MAGPHA 1
#define UIR_DFT_EPI_CABS 2
#define UIR_DFT_EPI_POLAR_BWD 3
typedef struct UirDft2Args {
  const float* in_re; const float* in_im;   /* in_im may be NULL (real input)                    */
  float* out_re; float* out_im;             /* out_im may be NULL (only the real part wanted)    */
  const float* ta; const float* tb;
  int O, N, I, C;                            /* tensor viewed as [O][N][I][C], transform along N  */
  float scale;
  int pro, epi;                              /* UIR_DFT_PRO_* / UIR_DFT_EPI_* (at most one of them non-zero)          */
  const float* aux0; const float* aux1; const float* aux2;   /* element-wise operands, same [O][N][I][C] layout       */
  float* aux_out;                            /* UIR_DFT_EPI_CABS: |out|                                               */
} UirDft2Args;
int uir_dft_axis2(const UirDft2Args* args, void* stream);

/* mag = |F|, pha = atan2(im, re) with Im forced to +0 at the self-conjugate bins (what
 * torch.fft.fft2 of a real tensor yields; tests/golden/fft_zero_sign.npz).  :177-178 */
int uir_magpha_fwd(const float* re, const float* im, float* mag, float* pha, int B, int H, int Wd, int C, void* stream);
/* (g_mag, g_pha) -> (g_re, g_im); zero where mag == 0 */
int uir_magpha_bwd(const float* mag, const float* pha, const float* g_mag, const float* g_pha,
                   float* g_re, float* g_im, long n, void* stream);
/* z = Mag * exp(i Pha)  (:183-185) and its adjoint */
int uir_polar_fwd(const float* mag, const float* pha, float* zr, float* zi, long n, void* stream);
int uir_polar_bwd(const float* mag, const float* pha, const float* g_zr, const float* g_zi,
                  float* g_mag, float* g_pha, long n, void* stream);
/* t = |z| (:187) and adjoint (sgn(0) = 0) */
int uir_cabs_fwd(const float* zr, const float* zi, float* t, long n, void* stream);
int uir_cabs_bwd(const float* zr, const float* zi, const float* t, const float* g_t, float* g_zr, float* g_zi, long n, void* stream);

/* ------------------------------------------------------------- uir_shuffle */
/* dst[b, 2h+dy, 2w+dx, dcol + c] = src[b, h, w, c] at (dy,dx)=(0,0), fill[c] elsewhere:
 * the zero-interleaved UpSample output after `post` (bias at the 3/4 empty pixels), :188. */
int uir_scatter_even(const float* src, int lds, const float* fill, float* dst, int ldd, int dcol,
                     int B, int H, int Wd, int C, void* stream);
int uir_gather_even(const float* src, int lds, int scol, float* dst, int ldd, int B, int H, int Wd, int C, void* stream);
/* PixelShuffle(2) / PixelUnshuffle(2) on HWC rows (:155, :195).  shuffle: src [B,H,W,4*C] ->
 * dst [B,2H,2W,C] (channel c*4 + dy*2 + dx -> pixel (2h+dy, 2w+dx)); unshuffle is the inverse. */
int uir_pixel_shuffle(const float* src, int lds, float* dst, int ldd, int dcol, int B, int H, int Wd, int C, void* stream);
int uir_pixel_unshuffle(const float* src, int lds, int scol, float* dst, int ldd, int B, int H, int Wd, int C, void* stream);
/* Patches of Conv2d(kernel 4, stride 2, padding 1) (Downsample, src/Models/AST.py:408-424) in the weight's own k order:
 * P[(b,oy,ox)][c*16 + ky*4 + kx] = x[(b, 2oy-1+ky, 2ox-1+kx)][c], 0 outside; x is [B, 2H, 2Wd, C] rows, P [B*H*Wd, 16*C]
 * (H, Wd: OUTPUT sides), so the convolution is one GEMM against weight.view(Cout, Cin*16).  _bwd is the adjoint
 * (dx[pixel][c] = the <= 4 patch entries holding it, summed in (ky, kx) order). */
int uir_patch4s2(const float* x, int ldx, float* P, int ldp, int B, int H, int Wd, int C, void* stream);
int uir_patch4s2_bwd(const float* dP, int ldp, float* dx, int ldx, int B, int H, int Wd, int C, void* stream);
/* Patches of a 3 x 3, padding 1 convolution in the weight's own k order: P[(b,y,x)][c*9 + ky*3 + kx] = x[(b, y-1+ky, x-1+kx)][c],
 * 0 outside (x [B*H*Wd, C] rows, P [B*H*Wd, 9*C]); _bwd is the adjoint.  The wide convolutions on few pixels (samplers of
 * src/model/block.py:107-153 at the deep levels, the VGG16 stacks of src/Losses/losses.py:215-250) run as one K = 9 C GEMM on them. */
int uir_patch3x3(const float* x, int ldx, float* P, int ldp, int B, int H, int Wd, int C, void* stream);
int uir_patch3x3_bwd(const float* dP, int ldp, float* dx, int ldx, int B, int H, int Wd, int C, void* stream);
/* dst[i] = sum_{g < group} src[idx[i * group + g]], entries with idx < 0 count as 0 (i < n; idx is device int32).  Weight
 * re-layouts of the AST samplers (src/Models/AST.py:408-443): the 4 x 4 stride-2 kernel of Downsample laid out as a 3 x 3
 * kernel over the 2 x 2-unshuffled image (and its gradient gathered back), the per-channel bias of the ConvTranspose2d of
 * Upsample repeated over the 2 x 2 output phases (and its gradient summed over them). */
int uir_gather_sum(const float* src, const int* idx, float* dst, long n, int group, void* stream);
/* dst[m][dcol + c] = src[m][scol + c] (+ dst if accumulate), c < C: torch.cat joins (:209,264-267) */
int uir_copy_cols(const float* src, int lds, int scol, float* dst, int ldd, int dcol, long M, int C, int accumulate, void* stream);
/* dst[0 .. n) = src[0 .. n): n % 4 == 0, both 16-byte aligned; an ordinary kernel in stream order (no blit / barrier packets around it) */
int uir_copy_flat(const float* src, float* dst, long n, void* stream);
/* slab[block][c] = sum over that block's rows of src[m][col + c]: bias gradients of the 1x1
 * convs with bias (SpectralTransformer.py:165-172) */
int uir_colsum_partials(const float* src, int ld, int col, long M, int C, float* slab, int nblocks, void* stream);
/* NCHW [B,3,H,W] <-> HWC rows [B*H*W][4] (4th column zero) at the model boundary */
int uir_nchw3_to_hwc4(const float* src, float* dst, int B, int H, int Wd, void* stream);
/* 3x3 patches (pad 1) of the NCHW [B,3,H,W] input as rows [B*H*W][28]: column cin*9 + ky*3 + kx, column 27 zero.
 * embed_conv_rgb (SpectralTransformer.py:219,254) and its weight gradient become a K = 27 uir_gemm / uir_gram. */
int uir_im2col3x3_c3(const float* src, float* dst, int B, int H, int Wd, void* stream);
int uir_hwc4_to_nchw3(const float* src, float* dst, int B, int H, int Wd, void* stream);

/* ---------------------------------------------------------------- uir_loss */
enum { UIR_LOSS_L1 = 0, UIR_LOSS_L2 = 1, UIR_LOSS_CHARBONNIER = 2, UIR_LOSS_LUMINANCE = 3, UIR_LOSS_L1_COLOR = 4,
       UIR_LOSS_LUM_CHARBONNIER = 5, /* "LuminanceCharbonnier" = LuminanceLoss + CharbonnierLoss, losses.py:118-123 */
       UIR_LOSS_ADAPTIVE_LUM_CHARBONNIER = 6, /* "AdaptiveLuminanceCharbonnier", losses.py:130-140: 0.5 * softmax(-log(losses + 1e-6))
                                               * weights (detached) over (Charbonnier, Luminance) */
       UIR_LOSS_GRADIENT = 7, /* "gradient" = Gradient_Loss, losses.py:162-181: L1 between the valid 3x3 Laplacians of pred and truth */
       /* with FocalFrequencyLoss(loss_weight 1, alpha 1) terms (losses.py:45-46; the focal_frequency_loss package is not vendored and
        * not in the image: its published algorithm - v0.3.0, patch_factor 1, ave_spectrum / log_matrix / batch_matrix False - is
        * restated by the uir_ffl_* kernels below, PARITY UNPINNED); the FFL mean is read from sums[6] */
       UIR_LOSS_FFL_CHARBONNIER = 8,      /* "fflCharbonnier" = ffl + charbonnier (:97-98) */
       UIR_LOSS_LUM_FFL = 9,              /* "LuminanceFFL" = ffl + luminance (:148-151) */
       UIR_LOSS_ADAPTIVE_LUM_FFL = 10,    /* "AdaptiveLuminanceFFL" (:137-147): the adaptive pair of kind 6 with ffl in place of charbonnier */
       /* with MS_SSIM(win 11, sigma 1.5, data_range 1, size_average, channel 3) (losses.py:43-44; pytorch_msssim is not vendored and not
        * in the image: restated by the uir_msssim_* functions below, PARITY UNPINNED); the MS-SSIM mean is read from sums[7] */
       UIR_LOSS_MS_SSIM = 11,             /* "ms_ssim" = 1 - ms_ssim (:86-87) */
       UIR_LOSS_SSIM_FFL_CHARBONNIER = 12 /* "ssimFFLCharbonnier" = 0.025 ffl + 0.025 charbonnier + (1 - ms_ssim) (:152-156) */ };
/* Partial sums over NCHW [B,3,H,W] pred/truth: slab[block][8] =
 * (sum|d|, sum d^2, sum sqrt(d^2+1e-6), sum lum^2, sum (clamp01(p)-clamp01(t))^2, 0,0,0).
 * Replaces nn.L1Loss / MSELoss / CharbonnierLoss / ColorLoss / LuminanceLoss
 * (src/Losses/losses.py:55-57,76-78,190-192,209-212; luminanceLoss.py:16-19) and the PSNR
 * reduction (src/ModelTrainer.py:17-21). */
int uir_loss_partials(const float* pred, const float* truth, float* slab, int nblocks, int B, int HW, void* stream);
/* sums[8] (reduced slab) -> out[4]: out[0] = loss(kind), out[1] = PSNR, out[2..3] = the two detached weights of
 * UIR_LOSS_ADAPTIVE_LUM_CHARBONNIER (0 otherwise).  UIR_LOSS_GRADIENT reads sums[5] = its mean (uir_gradloss_partials + reduction). */
int uir_loss_finalize(const float* sums, float* out, int kind, int B, int HW, void* stream);
/* dpred = dL/dpred * gscale[0]  (gscale = upstream scalar gradient on device); aux = the out[4] of uir_loss_finalize of the same
 * pair (read by UIR_LOSS_ADAPTIVE_LUM_CHARBONNIER only, may be NULL otherwise).  UIR_LOSS_GRADIENT: uir_gradloss_bwd. */
int uir_loss_bwd(const float* pred, const float* truth, const float* gscale, const float* aux, float* dpred, int kind, int B, int HW,
                 void* stream);
/* Gradient_Loss (src/Losses/losses.py:162-181): F.conv2d with the 3x3 Laplacian [[0,1,0],[1,-4,1],[0,1,0]] per channel, no padding
 * (output (H-2) x (W-2)), nn.L1Loss between the two results = mean |lap(pred - truth)|.  slab[block] = partial sum of |lap(d)|
 * (sum the nblocks entries and divide by B*3*(H-2)*(W-2)); H, W >= 3. */
int uir_gradloss_partials(const float* pred, const float* truth, float* slab, int nblocks, int B, int H, int Wd, void* stream);
/* dpred = gscale[0] / (B*3*(H-2)*(W-2)) * lap^T(sign(lap(pred - truth))) */
int uir_gradloss_bwd(const float* pred, const float* truth, const float* gscale, float* dpred, int B, int H, int Wd, void* stream);
/* FocalFrequencyLoss pieces around the 2-D DFT (which the host runs with uir_dft_axis* on HWC4 rows; ortho-normalised):
 *   uir_ffl_diff           d4[b][p][c] = pred[b][c][p] - truth[b][c][p] (c < 3), 0 in the pad channel: NCHW3 -> HWC4
 *   uir_ffl_max            mx2[b][c] = max_p |D[b][p][c]|^2 (slab: [B][nblk][4] scratch)
 *   uir_ffl_partials       slab[b * nblk + k] = partial sums of w |D|^2, w = |D| / sqrt(mx2) (0 where mx2 = 0: the package's
 *                          NaN -> 0), i.e. FFL = sum / (B 3 HW) - the spectrum weight matrix is detached in the package
 *   uir_ffl_spectrum_grad  G = coef * gscale[0] * (wdev ? wdev[0] : 1) * w * conj(D): its DFT (same sign) real part is dFFL/dd
 *   uir_ffl_add_grad       dpred[b][c][p] (+)= dd4[b][p][c]: HWC4 -> NCHW3 */
int uir_ffl_diff(const float* pred, const float* truth, float* d4, int B, int HW, void* stream);
int uir_ffl_max(const float* Dr, const float* Di, float* mx2, float* slab, int nblk, int B, int HW, void* stream);
int uir_ffl_partials(const float* Dr, const float* Di, const float* mx2, float* slab, int nblk, int B, int HW, void* stream);
int uir_ffl_spectrum_grad(const float* Dr, const float* Di, const float* mx2, const float* gscale, const float* wdev, float coef, float* Gr,
                          float* Gi, int B, int HW, void* stream);
int uir_ffl_add_grad(const float* dd4, float* dpred, int B, int HW, int accumulate, void* stream);

/* MS-SSIM as a training loss (pytorch_msssim.MS_SSIM restated: parity unpinned), NCHW planes [planes = B * 3][H][W]:
 *   uir_msssim_level      means[plane] = (mean ssim_map, mean cs_map) of one level: the 11-tap Gaussian as its outer-product window,
 *                         VALID positions only (slab: [planes][nblk][2] scratch)
 *   uir_avgpool2(_bwd)    F.avg_pool2d(x, 2, padding = size % 2) between levels (zeros counted) and its adjoint
 *   uir_msssim_combine    ms_out[0] = mean_planes prod_i relu(v_i)^w_i, v_i = cs mean of level i < levels - 1, ssim mean of the last;
 *                         dmeans[level][plane][2] = d ms / d(those means), 0 for the unused mean of each level
 *   uir_msssim_level_bwd  dX (+)= gradient of one level through its ssim / cs maps (A, Bm, Cm: [planes][(H-10)(W-10)] scratch)
 *   uir_axpy_dev          y += coef * gscale[0] * x */
typedef struct UirMsSsimArgs {
  const float* X; const float* Y;   /* [planes][H][W] */
  float* slab;                      /* uir_msssim_level only */
  int planes, H, W, nblk;
  float C1, C2;                     /* (0.01 * data_range)^2, (0.03 * data_range)^2 */
  float win[11];                    /* _fspecial_gauss_1d(11, 1.5), fp32 */
} UirMsSsimArgs;
typedef struct UirMsSsimWeights { float w[5]; } UirMsSsimWeights;
int uir_msssim_level(const UirMsSsimArgs* args, float* means, void* stream);
int uir_msssim_combine(const float* means, float* dmeans, float* ms_out, int levels, int planes, const UirMsSsimWeights* w, void* stream);
int uir_msssim_level_bwd(const UirMsSsimArgs* args, const float* dmeans, float* A, float* Bm, float* Cm, float* dX, int accumulate, void* stream);
int uir_avgpool2(const float* X, float* Y, int planes, int H, int W, void* stream);
int uir_avgpool2_bwd(const float* dY, float* dX, int planes, int H, int W, int accumulate, void* stream);
int uir_axpy_dev(const float* x, const float* gscale, float coef, float* y, long n, void* stream);

/* Pieces of VGGPerceptualLoss (src/Losses/losses.py:215-255) and TVLoss (:11-28) around the convolution stack, which is uir_gemm
 * (taps = 9, bias, ReLU = UIR_ACT_LEAKY with slope 0) / uir_leaky_bwd + uir_gemm (flipped taps) on HWC rows.  The VGG16 weights are
 * the USER's torchvision-format file (the reference downloads them; this build cannot and does not ship or fake them).
 *   uir_vgg_prep(_bwd)         out4[b][oy][ox][c] = (bilinear_{align_corners=False}(img)[b][c][oy][ox] - mean[c]) / std[c], S x S, pad
 *                              channel 0;  backward: dimg += coef * gscale[0] * adjoint(d4), gathered per source pixel (no atomics)
 *   uir_maxpool2_hwc(_bwd)     nn.MaxPool2d(2, 2) on rows [B*H*W][C]; gradient to the first maximum of each window in scan order
 *   uir_l1_feat_partials/_bwd  partial sums of |x - y| over n floats (n % 4 == 0); dX (+)= k * sign(x - y)
 *   uir_tv_partials/_bwd       slab[blk] = (sum of squared vertical, horizontal differences) of [planes][H][W];
 *                              dX += gscale[0] * (kh * d(sum_h)/dX + kw * d(sum_w)/dX) */
int uir_vgg_prep(const float* img, float* out4, int B, int H, int W, int S, void* stream);
int uir_vgg_prep_bwd(const float* d4, float* dimg, const float* gscale, float coef, int B, int H, int W, int S, void* stream);
int uir_maxpool2_hwc(const float* X, float* Y, int B, int H, int W, int C, void* stream);
int uir_maxpool2_hwc_bwd(const float* X, const float* dY, float* dX, int B, int H, int W, int C, int accumulate, void* stream);
int uir_l1_feat_partials(const float* X, const float* Y, float* slab, int nblocks, long n, void* stream);
int uir_l1_feat_bwd(const float* X, const float* Y, float* dX, float k, long n, int accumulate, void* stream);
int uir_tv_partials(const float* X, float* slab, int nblocks, int planes, int H, int W, void* stream);
int uir_tv_bwd(const float* X, float* dX, const float* gscale, float kh, float kw, int planes, int H, int W, void* stream);
int uir_relu_inplace(float* x, long n, void* stream);   /* x = max(x, 0), n % 4 == 0 */

/* --------------------------------------------------------------- uir_optim */
/* slab[block] = partial sum of squares of g[0..n) */
int uir_sumsq_partials(const float* g, long n, float* slab, int nblocks, void* stream);
/* clip_grad_norm_(max_norm) + Adam (src/ModelTrainer.py:87-88,199) over one flat buffer.
 * sumsq[0] = sum of squares of the (unscaled) gradients; gscale multiplies every gradient
 * first (1/world_size under data parallelism).  norm_out[0] receives the total norm. */
typedef struct UirAdamArgs {
  float* p; const float* g; float* m; float* v;
  const float* sumsq; float* norm_out;
  const float* hyper;  /* NULL in the eager path (the by-value fields below are used: kernel arguments are copied at launch,
                        * so an un-synchronised host can never overwrite a queued step's scalars).  hipGraph replay only:
                        * device float[4] = {lr, bc1, bc2, gscale}, stored by uir_adam_set_hyper on the replay stream. */
  long n;
  float lr, beta1, beta2, eps, bc1, bc2, max_norm, gscale;
  float weight_decay;  /* 0 = torch.optim.Adam (ModelTrainer.py:199); > 0 = torch.optim.AdamW's decoupled decay (:201) */
  int _pad;
} UirAdamArgs;
int uir_clip_adam_step(const UirAdamArgs* args, void* stream);
/* hyper[0..4) = {lr, bc1, bc2, gscale}, passed BY VALUE and written by a one-thread kernel in stream order */
int uir_adam_set_hyper(float* hyper, float lr, float bc1, float bc2, float gscale, void* stream);

/* -------------------------------------------------------------- uir_resize */
/* Input stage.  Replaces transforms.Resize((1024,1024), BILINEAR) -> ToTensor ->
 * Normalize(0.5, 0.5) of src/DataManipulation/DataLoader.py:13-20 (applied per item, UIEBDataset.py:30-32):
 * Pillow's 8-bit two-pass fixed-point resampler (horizontal, then vertical, 22-bit weights, uint8 intermediate)
 * followed by a 256-entry byte -> fp32 table.  Integer work: results are bit-identical to PIL.Image.resize.
 *
 * uir_resize_ksize / uir_resize_coeffs are HOST functions (no GPU): the window table bounds[out][2] = (first
 * source index, tap count) and the weights kk[out][ksize] of one axis, computed in double precision in the order
 * of Pillow's precompute_coeffs + normalize_coeffs_8bpc.  coeffs returns ksize, or -1 on a bad argument or when
 * out_size*ksize > kk_cap. */
int uir_resize_ksize(int in_size, int out_size);
int uir_resize_coeffs(int in_size, int out_size, int* bounds, int* kk, long kk_cap);
#define UIR_RESIZE_SRC_SLACK 16
typedef struct UirResizeImage {
  long src_off;        /* byte offset of the image in `src`: packed RGB, 3 bytes per pixel, rows of in_w pixels  */
  int in_h, in_w;
  int kh, kv;          /* taps per output column / row (the ksize of each axis table)                            */
  int hb, hk;          /* offsets (ints) into `tab`: horizontal bounds [out_w][2], weights [out_w][kh]           */
  int vb, vk;          /* vertical bounds [out_h][2], weights [out_h][kv]                                        */
} UirResizeImage;
typedef struct UirResizeArgs {
  const unsigned char* src;        /* DEVICE: all source images of the batch, back to back                        */
  long src_bytes;                  /* bytes the images occupy                                                     */
  long src_capacity;               /* readable bytes at src: >= src_bytes + UIR_RESIZE_SRC_SLACK (windows are     */
                                   /* fetched as 12-byte runs; the bytes past a window never reach the result)    */
  const UirResizeImage* img;       /* DEVICE [n_img]                                                              */
  const UirResizeImage* img_host;  /* HOST copy of img: argument checks (every window inside its image)            */
  const int* tab;                  /* DEVICE: the axis tables the descriptors point into                          */
  const int* tab_host;             /* HOST copy of tab                                                            */
  long tab_ints;
  const float* lut;                /* DEVICE [256]: fp32 value of each byte after ToTensor + Normalize             */
  float* out_f32;                  /* DEVICE [n_img][3][out_h][out_w] (the loader's item layout) or NULL          */
  unsigned char* out_u8;           /* DEVICE [n_img][out_h][out_w][3]: the resampled bytes, or NULL               */
  int n_img, out_h, out_w;
} UirResizeArgs;
int uir_resize_norm_u8(const UirResizeArgs* args, void* stream);

/* ---------------------------------------------------------------- uir_ssim */
/* Validation SSIM: torchSSIM(ref, out) = pytorch_msssim.ssim(tar, prd, data_range=1.0, size_average=True)
 * (src/ModelTrainer.py:16,23-24,149,156).  The package is not vendored in the reference: its published algorithm is
 * restated (11-tap Gaussian sigma 1.5, separable VALID convolution per channel, K = (0.01, 0.03)); PARITY UNPINNED against
 * the package itself.  One slab entry per 16 x 32 tile of the valid output of each (sample, channel) plane = the sum of
 * ssim_map over the tile; SSIM = uir_reduce_slab(slab, nblocks, 1, out, 1, 0, 1 / (planes (H-10) (W-10))).
 * uir_ssim_nblocks is a host helper (returns -1 if H or W < 11). */
#define UIR_SSIM_TILE_H 16
#define UIR_SSIM_TILE_W 32
typedef struct UirSsimArgs {
  const float* X; const float* Y;   /* [planes][H][W] each (planes = B * C of two NCHW tensors)                */
  float* slab; long slab_floats;    /* >= uir_ssim_nblocks(planes, H, W)                                       */
  int planes, H, W;
  float C1, C2;                     /* (0.01 * data_range)^2, (0.03 * data_range)^2                            */
  float win[11];                    /* the normalised Gaussian window (fp32, as _fspecial_gauss_1d(11, 1.5))   */
} UirSsimArgs;
long uir_ssim_nblocks(int planes, int H, int W);
int uir_ssim_partials(const UirSsimArgs* args, void* stream);

/* ---------------------------------------------------------------- uir_frfn */
/* FRFN feed-forward of the NewBigFRFNModel family: src/model/block.py:249-282.  Its contractions are
 * uir_gemm (Linear C -> 8C with bias :252,273; Linear 4C -> C with bias :254,281; partial 3x3 convolution on the first C/4
 * channels :259,267-269), uir_dwconv / uir_dwconv_bwd_fused (depthwise 3x3 of the x1 half :253,277) and uir_gram /
 * uir_colsum_partials (weight and bias gradients).  These four are the element-wise steps in between:
 *   t = GELU(pre)                    (:252 nn.GELU after linear1)          uir_frfn_act_fwd / _bwd (dpre = dt * GELU'(pre))
 *   g = GELU(u + bias) * v           (:253 conv bias + nn.GELU, :279 gate) uir_frfn_gate_fwd / _bwd (dU, dV from dG)
 * rows = tokens (pixels), `cols` / `hid` channels (multiples of 4), ld* = row strides in floats. */
int uir_frfn_act_fwd(const float* pre, int ldp, float* t, int ldt, long M, int cols, void* stream);
/* db_slab != NULL: also per-workgroup column sums of dpre (= the bias gradient of linear1), slab[nblocks][cols], the launch
 * uses exactly nblocks workgroups; needs 256 % (cols/4) == 0.  Summed by uir_reduce_slab(_batch). */
int uir_frfn_act_bwd(const float* pre, int ldp, const float* dt, int ldt, float* dpre, int ldo, long M, int cols, float* db_slab,
                     int nblocks, void* stream);
/* dz = dy * (y > 0 ? 1 : slope): backward of LeakyReLU from its output (InputProjection's nn.LeakyReLU(inplace=True), block.py:50) */
int uir_leaky_bwd(const float* y, int ldy, const float* dy, int lddy, float* dz, int lddz, long M, int cols, float slope, void* stream);
typedef struct UirFrfnGateArgs {
  const float* U;      /* depthwise-conv output without bias [M][ldu] */
  const float* bias;   /* [hid] */
  const float* V;      /* the x2 half [M][ldv]; NULL: no gate (v = 1), G = GELU(U + bias) - LeFF's depthwise stage (block.py:228,240) */
  float* G;            /* forward: out [M][ldg] */
  const float* dG;     /* backward: gradient of G [M][ldg] */
  float* dU;
  float* dV;           /* NULL iff V is NULL */
  float* db_slab;      /* backward, optional: per-workgroup column sums of dU (= the depthwise bias gradient), [db_blocks][hid] */
  long M;
  int hid;
  int ldu, ldv, ldg, lddu, lddv;
  int db_blocks;       /* workgroups of the backward launch when db_slab is given (256 % (hid/4) == 0 required) */
  int v_act;           /* 1: V holds the PRE-activation of the x2 half (Linear1's output, block.py:271-273): the gate uses gelu(V), and
                          dV is the gradient with respect to that pre-activation (dG gelu(U + bias) gelu'(V)) - no activated copy of
                          Linear1's output is stored and no separate GELU-backward pass runs */
  float* dbv_slab;     /* backward, optional, with db_slab: per-workgroup column sums of dV (Linear1's bias gradient, x2 half) */
} UirFrfnGateArgs;
int uir_frfn_gate_fwd(const UirFrfnGateArgs* args, void* stream);
int uir_frfn_gate_bwd(const UirFrfnGateArgs* args, void* stream);

/* ------------------------------------------------------------ token-axis ops */
/* EncoderBlock of the NewBigFRFNModel family (src/model/model.py:57-92).
 * uir_layernorm_fwd: nn.LayerNorm(C) over token rows (:62-63), Y = (X - mean) * rstd * gamma + beta, eps inside the root;
 *   stats[m] = (mean, rstd) as uir_layernorm_bwd expects them; C in {16, 32, 64, 128}.
 * fft.fftn(x, dim=(-2,-1)).real (:72) and fft.ifftn(...).real (:86) over the (L, C) axes of (B, L, C) tokens are one real-
 * symmetric operator: two uir_gemm against cos / sin matrices for C, and for L = N1 * N2 a two-stage
 * transform on uir_dft_axis with
 *   uir_cmul_table      (re, im)[r][c] *= (tab_re, tab_im)[r % P]: the inter-stage twiddle exp(2 pi i k1 n2 / L), P = N1 * N2
 *   uir_transpose_rows  dst[b][k3][k2][k1][:] = scale * src[b][k1][k2][k3][:]: the stage-ordered frequency index
 *                       k1 + N1 (k2 + N2 k3) back to natural token order (N3 = 1 or N2 = 1: the two-stage transpose). */
int uir_layernorm_fwd(const float* X, int ldx, const float* gamma, const float* beta, float* Y, int ldy, float* stats, long M, int C,
                      float eps, void* stream);
int uir_cmul_table(float* re, float* im, const float* tab_re, const float* tab_im, long rows, long P, int C, void* stream);
int uir_transpose_rows(const float* src, float* dst, int B, int N1, int N2, int N3, int C, float scale, void* stream);
/* dst[m][:] (+)= scale[m / rows_per_sample] * src[m][:] - DropPath (timm drop_path: one Bernoulli(keep) / keep factor per sample;
 * EncoderBlock, model.py:27,54,91); scale is a device float[M / rows_per_sample]. */
int uir_scale_rows(const float* src, int lds, float* dst, int ldd, const float* scale, long rows_per_sample, long M, int C, int accumulate,
                   void* stream);
/* dst[m][:] = a[m][:] + scale[m / rows_per_sample] * b[m][:] (scale NULL: a + b): a residual join in one pass - `x + drop_path(f(x))`
 * of AST's TransformerBlock (src/Models/AST.py:617,620), the shortcut sums of MDASSA / DecoderBlock (src/model/block.py:465,514,
 * src/model/model.py:152,156).  FMA per element: (a + s b) rounded once. */
int uir_add_scaled_rows(const float* a, int lda, const float* b, int ldb, float* dst, int ldd, const float* scale, long rows_per_sample,
                        long M, int C, void* stream);
/* Last stage of a length-N = 2 N2 complex DFT (sign +) along an axis of rows [O][N][I][C] whose first stage transformed the two
 * interleaved sub-sequences n = 2 n2 + n1 (uir_dft_axis2 over n2 with inner extent 2 I): Y rows [o][k2][n1][i], twiddle table
 * exp(2 pi i k2 / N) [N2];  X[o][k1 N2 + k2][i] = scale (Y[k2][0] + (-1)^k1 tw[k2] Y[k2][1]), natural order; x_im NULL = real part
 * only.  Used for the 256-long axes of FDFP's fft2 (block.py:538,549). */
int uir_radix2_combine(const float* y_re, const float* y_im, float* x_re, float* x_im, const float* tab_re, const float* tab_im, long O,
                       int N2, long I, int C, float scale, void* stream);

/* -------------------------------------------------------------- uir_winattn */
/* Attention core of WindowAttention_Sparse (src/model/block.py:336-370) on 8 x 8 non-shifted windows of an HWC token image
 * (what every DecoderBlock / MDASSA of NewBigFRFNModel uses: win_size 8, shift_size 0, mask None):
 *   attn = (Q scale) K^T + table[relative_position_index]; attn = softmax(attn) w1 + relu(attn)^2 w2, (w1, w2) = softmax(w);
 *   O = attn V, heads along the channels (head h = columns [h C/heads, (h+1) C/heads)).
 * Q, K, V, O are token rows (the outputs of to_q / to_kv / the input of proj, which are uir_gemm calls).  Backward: dQ, dK, dV rows
 * from dO, plus per-window partials dtable_slab[window][(2 ws - 1)^2 * heads] (same layout as the table) and
 * dw_slab[window * heads + h][2] = gradients w.r.t. (w1, w2); uir_winattn_wgrad maps their sum to the gradient of `w`.
 * C / heads in {8, 16, 32, 64, 128} (walked in chunks of <= 32 channels); ws = 8; H, W multiples of 8. */
typedef struct UirWinAttnArgs {
  const float* Q; const float* K; const float* V;
  const float* table;      /* relative_position_bias_table [(2 ws - 1)^2][heads] */
  const float* w;          /* [2] */
  float* O;
  const float* dO;
  float* dQ; float* dK; float* dV;
  float* dtable_slab;
  float* dw_slab;
  int ldq, ldk, ldv, ldo, lddq, lddk, lddv;
  int B, H, W, C, heads, ws;
  float scale;             /* qk_scale or head_dim^-0.5 (:291) */
  int shift;               /* 0: plain windows.  s in (0, ws): the shifted windows of AST's TransformerBlock (src/Models/AST.py:568-616):
                              window token (ty, tx) of window (wy, wx) is image pixel ((wy ws + ty + s) mod H, (wx ws + tx + s) mod W)
                              (torch.roll by -s, window_partition, window_reverse, roll by +s - no data moves), and the additive
                              -100 mask between tokens of different shift regions (:568-589) is applied to the scores */
} UirWinAttnArgs;
int uir_winattn_fwd(const UirWinAttnArgs* args, void* stream);
int uir_winattn_bwd(const UirWinAttnArgs* args, void* stream);
int uir_winattn_wgrad(const float* w, const float* dsum, float* dw, void* stream);

/* ---------------------------------------------------------------- uir_uiqm */
/* Evaluation metric UIQM.  Replaces getUIQM of uqim_utils.py:176-190 (with _uicm :36-48, mu_a :10-28,
 * s_a :30-34, sobel :50-54, eme :57-86, _uism :88-113, _uiconm :141-174) for a batch of N images of one size, HWC, either
 * float32 (the reference's x.astype(np.float32)) or the uint8 bytes eval.ipynb reads (cv2.imread + resize).
 * out[n] = (uiqm, uicm, uism, uiconm) in double, like the reference's Python floats.  flags[n] != 0: a window's max/min ratio
 * was <= 0 (signed image) - the reference raises ValueError("math domain error") there; the host mirror does the same.
 * work: uir_uiqm_work_doubles(N, H, W) doubles of scratch (-1 if H or W < 10, one 10 x 10 window). */
typedef struct UirUiqmArgs {
  const float* img_f32;
  const unsigned char* img_u8;   /* exactly one of img_f32 / img_u8 is non-NULL */
  double* work;
  double* out;                   /* [N][4] */
  int* flags;                    /* [N] */
  int N, H, W;
} UirUiqmArgs;
long uir_uiqm_work_doubles(int N, int H, int W);
int uir_uiqm(const UirUiqmArgs* args, void* stream);

/* Host helpers: a stream restricted to n_enabled compute units (hipExtStreamCreateWithCUMask; the driver spreads the enabled
 * bits over the 8 XCDs), for the weight-gradient lane of the train step.  The caller owns the stream. */
int uir_stream_create_cu_masked(int n_enabled, void** out_stream);
int uir_stream_destroy(void* stream);
/* Ordering events between the streams of ONE device (the step's chain / side / bulk lanes): no timing, no system-scope fence at the record
 * (hipEventDisableTiming | hipEventDisableSystemFence) - a default event's record writes the caches back for the host and other devices,
 * which nothing here reads, and stalls the recording queue for it.  Not for host synchronisation. */
int uir_event_create(void** out_event);
int uir_event_destroy(void* event);
int uir_event_record(void* event, void* stream);
int uir_stream_wait_event(void* stream, void* event);

/* library self-description (used by the loader test) */
const char* uir_version(void);
/* hipGetErrorString of the HIP error behind this thread's most recent UIR_ELAUNCH ("no error" if there was none) */
const char* uir_last_launch_error(void);
/* Layout of an argument struct as the compiler built it: out[0] = sizeof, out[1 + i] = offsetof(field i) in declaration
 * order; returns 1 + number of fields, -1 for an unknown name / too small a buffer.  A host binding (ctypes, cgo, JNI)
 * checks its mirror of each struct against this at load time (csrc/abi_layout.hip is generated from this header). */
long uir_abi_layout(const char* struct_name, long* out, int cap);

#ifdef __cplusplus
}
#endif
#endif
