"""ctypes binding of libuir_hip.so (C ABI: include/uir_kernels.h).

There is NO fallback: if the shared library is missing or a symbol does not resolve, importing
this module raises.  The product path never routes through PyTorch eager ops or the oracle.
"""
from __future__ import annotations

import ctypes as C
import os

# torch FIRST: it brings its own HIP runtime (torch/lib/libamdhip64.so, soname libamdhip64.so.7) and libuir_hip.so must bind to THAT one -
# the streams and device pointers it is handed belong to it.  Loaded before torch, the library would pull in /opt/rocm's runtime instead,
# torch would then load its own beside it, and every launch would fail ("no ROCm-capable device is detected").
import torch  # noqa: F401

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libuir_hip.so")

fp = C.c_void_p  # device pointers travel as integers

# enums (mirror include/uir_kernels.h)
PRO_NONE, PRO_LN, PRO_GATE = 0, 1, 2
EPI_PLAIN, EPI_GATE_BWD, EPI_LEAKY_BWD, EPI_GATE_BWD_DW, EPI_GELU_BWD, EPI_PLAIN_DW, EPI_LN_BWD = 0, 1, 2, 3, 4, 5, 6
ACT_NONE, ACT_LEAKY, ACT_GELU = 0, 1, 2
# L1ColorLum has the same formula as L1withColor in the reference (src/Losses/losses.py:59-66 vs :67-75)
LOSS_KINDS = {"L1": 0, "L2": 1, "charbonnier": 2, "Luminance": 3, "L1withColor": 4, "L1ColorLum": 4, "LuminanceCharbonnier": 5,
              "AdaptiveLuminanceCharbonnier": 6, "gradient": 7,
              # with a FocalFrequencyLoss term (src/Losses/losses.py:45-46,97-98,137-151; the package's algorithm restated: parity unpinned)
              "fflCharbonnier": 8, "LuminanceFFL": 9, "AdaptiveLuminanceFFL": 10,
              # with an MS_SSIM term (:43-44,86-87,152-156; pytorch_msssim restated: parity unpinned)
              "ms_ssim": 11, "ssimFFLCharbonnier": 12,
              # with the VGG16 perceptual term (:82-83,88-117; composed on the host from the terms above, Losses/perceptual.py): they need the
              # USER's torchvision weight file and refuse without it
              "perceptual": 100, "mix": 101, "fflMixTV": 102, "bigMix": 103, "fflMix": 104}
LOSS_GRADIENT = 7
LOSS_FFL_KINDS = (8, 9, 10, 12)
LOSS_MSSSIM_KINDS = (11, 12)
LOSS_MIX_FIRST = 100             # kinds >= this are host-composed mixes
LOSS_ADAPTIVE_KINDS = (6, 10)      # term weights from the detached loss values: aux of loss_sums -> loss_grad


class GemmArgs(C.Structure):
    _fields_ = [("X", fp), ("W", fp), ("Y", fp), ("bias", fp), ("R", fp), ("ln_g", fp), ("ln_b", fp),
                ("stats", fp), ("G", fp),
                ("x_bs", C.c_long), ("y_bs", C.c_long), ("r_bs", C.c_long), ("g_bs", C.c_long), ("w_bs", C.c_long),
                ("w_sn", C.c_long), ("w_sk", C.c_long), ("w_st", C.c_long),
                ("ldx", C.c_int), ("ldy", C.c_int), ("ldr", C.c_int), ("ldg", C.c_int),
                ("M", C.c_int), ("N", C.c_int), ("K", C.c_int), ("batches", C.c_int),
                ("taps", C.c_int), ("flip", C.c_int), ("H", C.c_int), ("Wd", C.c_int),
                ("n_grp_real", C.c_int), ("n_grp_pad", C.c_int), ("k_grp_real", C.c_int), ("k_grp_pad", C.c_int),
                ("pro", C.c_int), ("gate_off", C.c_int), ("epi", C.c_int), ("g_off", C.c_int), ("act", C.c_int),
                ("slope", C.c_float), ("ln_eps", C.c_float), ("dw_slab", fp), ("dw_stride", C.c_long),
                ("tap0", C.c_int), ("ntap", C.c_int), ("pre_out", fp), ("ldpre", C.c_long),
                ("out_scale", fp), ("scale_rows", C.c_int), ("dw_cols", C.c_int), ("dw_rows", C.c_int), ("il16", C.c_int)]


class GramArgs(C.Structure):
    _fields_ = [("A", fp), ("B", fp), ("slab", fp), ("stats", fp), ("ln_g", fp), ("ln_b", fp),
                ("a_bs", C.c_long), ("b_bs", C.c_long),
                ("o_sn", C.c_long), ("o_sk", C.c_long), ("o_st", C.c_long), ("o_bs", C.c_long),
                ("slab_stride", C.c_long),
                ("sq_a_off", C.c_long), ("sq_b_off", C.c_long), ("cs_a_off", C.c_long),
                ("lda", C.c_int), ("ldb", C.c_int), ("M", C.c_int), ("N", C.c_int), ("K", C.c_int),
                ("batches", C.c_int),
                ("n_grp_real", C.c_int), ("n_grp_pad", C.c_int), ("k_grp_real", C.c_int), ("k_grp_pad", C.c_int),
                ("taps", C.c_int), ("H", C.c_int), ("Wd", C.c_int), ("pro_b", C.c_int), ("gate_off", C.c_int),
                ("rows_per_chunk", C.c_int), ("nchunks", C.c_int)]


class Conv3bArgs(C.Structure):
    _fields_ = [("X", fp), ("Wp", fp), ("Y", fp), ("R", fp), ("ldx", C.c_int), ("ldy", C.c_int), ("ldr", C.c_int),
                ("B", C.c_int), ("H", C.c_int), ("Wd", C.c_int), ("N", C.c_int), ("K", C.c_int), ("tile", C.c_int)]


class Conv3bWgradArgs(C.Structure):
    _fields_ = [("dY", fp), ("X", fp), ("slab", fp), ("slab_stride", C.c_long), ("ldy", C.c_int), ("ldx", C.c_int),
                ("B", C.c_int), ("H", C.c_int), ("Wd", C.c_int), ("N", C.c_int), ("K", C.c_int), ("nchunks", C.c_int)]


class AttnvBwdArgs(C.Structure):
    _fields_ = [("dX1", fp), ("WA", fp), ("WB", fp), ("Wdw", fp), ("U", fp), ("VF", fp), ("V", fp), ("dV", fp), ("slab_ab", fp), ("slab_dw", fp),
                ("slab_ab_stride", C.c_long), ("nab", C.c_long), ("ldx", C.c_int), ("ldu", C.c_int), ("ldf", C.c_int), ("ldv", C.c_int), ("ldd", C.c_int),
                ("B", C.c_int), ("H", C.c_int), ("Wd", C.c_int), ("C", C.c_int)]


class DwArgs(C.Structure):
    _fields_ = [("X", fp), ("Wt", fp), ("Y", fp), ("ldx", C.c_int), ("ldy", C.c_int),
                ("B", C.c_int), ("H", C.c_int), ("Wd", C.c_int), ("C", C.c_int),
                ("grp_real", C.c_int), ("grp_pad", C.c_int), ("flip", C.c_int),
                ("act_in", C.c_int), ("bias", fp), ("G", fp), ("ldg", C.c_int)]


class DwWgradArgs(C.Structure):
    _fields_ = [("dY", fp), ("X", fp), ("slab", fp), ("ldy", C.c_int), ("ldx", C.c_int),
                ("B", C.c_int), ("H", C.c_int), ("Wd", C.c_int), ("C", C.c_int),
                ("grp_real", C.c_int), ("grp_pad", C.c_int), ("creal", C.c_int), ("nblocks", C.c_int)]


class DwBwdArgs(C.Structure):
    _fields_ = [("dY", fp), ("X", fp), ("Wt", fp), ("dX", fp), ("slab", fp),
                ("ldy", C.c_int), ("ldx", C.c_int), ("ldo", C.c_int),
                ("B", C.c_int), ("H", C.c_int), ("Wd", C.c_int), ("C", C.c_int),
                ("grp_real", C.c_int), ("grp_pad", C.c_int), ("creal", C.c_int), ("nblocks", C.c_int),
                ("act", C.c_int), ("db_slab", fp), ("U", fp), ("gbias", fp), ("dbu_slab", fp), ("ldu", C.c_int)]


class GdfnFwdArgs(C.Structure):
    _fields_ = [("X", fp), ("ln_g", fp), ("ln_b", fp), ("Win", fp), ("Wdw", fp), ("Wout", fp),
                ("Y", fp), ("T1", fp), ("T2", fp), ("stats", fp),
                ("ldx", C.c_int), ("ldy", C.c_int), ("ldt", C.c_int),
                ("B", C.c_int), ("H", C.c_int), ("Wd", C.c_int), ("C", C.c_int),
                ("h", C.c_int), ("hp", C.c_int), ("ln_eps", C.c_float), ("il", C.c_int)]


class QkvFwdArgs(C.Structure):
    _fields_ = [("X", fp), ("ln_g", fp), ("ln_b", fp), ("Wqkv", fp), ("Wdw", fp), ("Ypre", fp), ("Y", fp), ("stats", fp),
                ("ldx", C.c_int), ("ldp", C.c_int), ("ldq", C.c_int),
                ("B", C.c_int), ("H", C.c_int), ("Wd", C.c_int), ("C", C.c_int), ("ln_eps", C.c_float),
                ("gram_slab", fp), ("gram_stride", C.c_long)]


class AttnvFwdArgs(C.Structure):
    _fields_ = [("V", fp), ("WA", fp), ("WB", fp), ("Wdw", fp), ("R", fp), ("U", fp), ("VF", fp), ("Y", fp),
                ("ldv", C.c_int), ("ldr", C.c_int), ("ldu", C.c_int), ("ldf", C.c_int), ("ldy", C.c_int),
                ("B", C.c_int), ("H", C.c_int), ("Wd", C.c_int), ("C", C.c_int)]


class DwLnBwdArgs(C.Structure):
    _fields_ = [("G", fp), ("XP", fp), ("Wdw", fp), ("W1", fp), ("X", fp), ("stats", fp), ("gamma", fp), ("beta", fp),
                ("dRes", fp), ("dX", fp), ("slab", fp), ("slab_stride", C.c_long),
                ("ldg", C.c_int), ("ldp", C.c_int), ("ldx", C.c_int), ("ldr", C.c_int), ("ldo", C.c_int),
                ("B", C.c_int), ("H", C.c_int), ("Wd", C.c_int), ("C", C.c_int), ("h", C.c_int), ("hp", C.c_int),
                ("ch", C.c_int), ("nblocks", C.c_int), ("il", C.c_int)]


class LnBwdArgs(C.Structure):
    _fields_ = [("dT", fp), ("X", fp), ("stats", fp), ("gamma", fp), ("dRes", fp), ("dX", fp), ("slab", fp),
                ("ldt", C.c_int), ("ldx", C.c_int), ("ldr", C.c_int), ("ldo", C.c_int),
                ("M", C.c_int), ("C", C.c_int), ("nblocks", C.c_int)]


class AttnFwdArgs(C.Structure):
    _fields_ = [("G", fp), ("sq_q", fp), ("sq_k", fp), ("temperature", fp), ("P", fp), ("Wof", fp),
                ("attn", fp), ("WA", fp), ("WB", fp), ("B", C.c_int), ("C", C.c_int), ("heads", C.c_int),
                ("slab", fp), ("nch", C.c_int), ("slab_stride", C.c_long), ("sq_q_off", C.c_long), ("sq_k_off", C.c_long)]


class AttnBwdArgs(C.Structure):
    _fields_ = [("dWA", fp), ("dWB", fp), ("G", fp), ("sq_q", fp), ("sq_k", fp), ("temperature", fp),
                ("attn", fp), ("P", fp), ("Wof", fp), ("Mqk", fp), ("dP", fp), ("dWof", fp), ("dT", fp),
                ("B", C.c_int), ("C", C.c_int), ("heads", C.c_int),
                ("slab", fp), ("nch", C.c_int), ("slab_stride", C.c_long), ("dWA_off", C.c_long), ("dWB_off", C.c_long)]


class DftArgs(C.Structure):
    _fields_ = [("in_re", fp), ("in_im", fp), ("out_re", fp), ("out_im", fp), ("cos_t", fp), ("sin_t", fp),
                ("O", C.c_int), ("N", C.c_int), ("I", C.c_int), ("C", C.c_int),
                ("sign", C.c_float), ("scale", C.c_float)]


class Dft2Args(C.Structure):
    _fields_ = [("in_re", fp), ("in_im", fp), ("out_re", fp), ("out_im", fp), ("ta", fp), ("tb", fp),
                ("O", C.c_int), ("N", C.c_int), ("I", C.c_int), ("C", C.c_int), ("scale", C.c_float),
                ("pro", C.c_int), ("epi", C.c_int), ("aux0", fp), ("aux1", fp), ("aux2", fp), ("aux_out", fp)]


DFT_PRO_POLAR, DFT_PRO_CABS_BWD, DFT_PRO_MAGPHA_BWD = 1, 2, 3
DFT_EPI_MAGPHA, DFT_EPI_CABS, DFT_EPI_POLAR_BWD = 1, 2, 3
RESIZE_SRC_SLACK = 16


class ResizeImage(C.Structure):
    _fields_ = [("src_off", C.c_long), ("in_h", C.c_int), ("in_w", C.c_int), ("kh", C.c_int), ("kv", C.c_int),
                ("hb", C.c_int), ("hk", C.c_int), ("vb", C.c_int), ("vk", C.c_int)]


class ResizeArgs(C.Structure):
    _fields_ = [("src", fp), ("src_bytes", C.c_long), ("src_capacity", C.c_long), ("img", fp), ("img_host", fp), ("tab", fp), ("tab_host", fp),
                ("tab_ints", C.c_long), ("lut", fp), ("out_f32", fp), ("out_u8", fp),
                ("n_img", C.c_int), ("out_h", C.c_int), ("out_w", C.c_int)]


class SsimArgs(C.Structure):
    _fields_ = [("X", fp), ("Y", fp), ("slab", fp), ("slab_floats", C.c_long), ("planes", C.c_int), ("H", C.c_int), ("W", C.c_int),
                ("C1", C.c_float), ("C2", C.c_float), ("win", C.c_float * 11)]


class ReduceDesc(C.Structure):
    _fields_ = [("slab", fp), ("out", fp), ("stride", C.c_long), ("n", C.c_long), ("nchunks", C.c_int),
                ("accumulate", C.c_int), ("scale", C.c_float), ("_pad", C.c_int)]


class FrfnGateArgs(C.Structure):
    _fields_ = [("U", fp), ("bias", fp), ("V", fp), ("G", fp), ("dG", fp), ("dU", fp), ("dV", fp), ("db_slab", fp), ("M", C.c_long),
                ("hid", C.c_int), ("ldu", C.c_int), ("ldv", C.c_int), ("ldg", C.c_int), ("lddu", C.c_int), ("lddv", C.c_int),
                ("db_blocks", C.c_int), ("v_act", C.c_int), ("dbv_slab", fp)]


class MsSsimArgs(C.Structure):
    _fields_ = [("X", fp), ("Y", fp), ("slab", fp), ("planes", C.c_int), ("H", C.c_int), ("W", C.c_int), ("nblk", C.c_int),
                ("C1", C.c_float), ("C2", C.c_float), ("win", C.c_float * 11)]


class MsSsimWeights(C.Structure):
    _fields_ = [("w", C.c_float * 5)]


class WinAttnArgs(C.Structure):
    _fields_ = [("Q", fp), ("K", fp), ("V", fp), ("table", fp), ("w", fp), ("O", fp), ("dO", fp), ("dQ", fp), ("dK", fp), ("dV", fp),
                ("dtable_slab", fp), ("dw_slab", fp),
                ("ldq", C.c_int), ("ldk", C.c_int), ("ldv", C.c_int), ("ldo", C.c_int), ("lddq", C.c_int), ("lddk", C.c_int),
                ("lddv", C.c_int), ("B", C.c_int), ("H", C.c_int), ("W", C.c_int), ("C", C.c_int), ("heads", C.c_int),
                ("ws", C.c_int), ("scale", C.c_float), ("shift", C.c_int)]


class UiqmArgs(C.Structure):
    _fields_ = [("img_f32", fp), ("img_u8", fp), ("work", fp), ("out", fp), ("flags", fp),
                ("N", C.c_int), ("H", C.c_int), ("W", C.c_int)]


class AdamArgs(C.Structure):
    _fields_ = [("p", fp), ("g", fp), ("m", fp), ("v", fp), ("sumsq", fp), ("norm_out", fp), ("hyper", fp), ("n", C.c_long),
                ("lr", C.c_float), ("beta1", C.c_float), ("beta2", C.c_float), ("eps", C.c_float),
                ("bc1", C.c_float), ("bc2", C.c_float), ("max_norm", C.c_float), ("gscale", C.c_float),
                ("weight_decay", C.c_float), ("_pad", C.c_int)]


i32, i64, f32 = C.c_int, C.c_long, C.c_float

# name -> argtypes (every symbol include/uir_kernels.h declares)
SIGNATURES = {
    "uir_gemm": [C.POINTER(GemmArgs), fp],
    "uir_gram": [C.POINTER(GramArgs), fp],
    "uir_reduce_slab": [fp, i32, i64, fp, i64, i32, f32, fp],
    "uir_reduce_slab_batch": [fp, fp, i32, i32, fp],
    "uir_conv3x3_wsplit": [fp, i32, i32, fp, fp, fp],
    "uir_conv3x3_split": [C.POINTER(Conv3bArgs), fp],
    "uir_conv3x3_split_wgrad": [C.POINTER(Conv3bWgradArgs), fp],
    "uir_dwconv": [C.POINTER(DwArgs), fp],
    "uir_dwconv_wgrad": [C.POINTER(DwWgradArgs), fp],
    "uir_dwconv_bwd_fused": [C.POINTER(DwBwdArgs), fp],
    "uir_gdfn_fwd": [C.POINTER(GdfnFwdArgs), fp],
    "uir_qkv_fwd": [C.POINTER(QkvFwdArgs), fp],
    "uir_attnv_fwd": [C.POINTER(AttnvFwdArgs), fp],
    "uir_dwln_bwd": [C.POINTER(DwLnBwdArgs), fp],
    "uir_attnv_bwd": [C.POINTER(AttnvBwdArgs), fp],
    "uir_attnv_bwd_rows": [i32, i32],                    # host helper: returns a count, not a status
    "uir_attnv_bwd_rows_c": [i32, i32, i32],             # the same for any supported C
    "uir_layernorm_bwd": [C.POINTER(LnBwdArgs), fp],
    "uir_attn_fwd": [C.POINTER(AttnFwdArgs), fp],
    "uir_attn_bwd": [C.POINTER(AttnBwdArgs), fp],
    "uir_dft_axis": [C.POINTER(DftArgs), fp],
    "uir_dft_axis2": [C.POINTER(Dft2Args), fp],
    "uir_magpha_fwd": [fp, fp, fp, fp, i32, i32, i32, i32, fp],
    "uir_magpha_bwd": [fp, fp, fp, fp, fp, fp, i64, fp],
    "uir_polar_fwd": [fp, fp, fp, fp, i64, fp],
    "uir_polar_bwd": [fp, fp, fp, fp, fp, fp, i64, fp],
    "uir_cabs_fwd": [fp, fp, fp, i64, fp],
    "uir_cabs_bwd": [fp, fp, fp, fp, fp, fp, i64, fp],
    "uir_scatter_even": [fp, i32, fp, fp, i32, i32, i32, i32, i32, i32, fp],
    "uir_gather_even": [fp, i32, i32, fp, i32, i32, i32, i32, i32, fp],
    "uir_pixel_shuffle": [fp, i32, fp, i32, i32, i32, i32, i32, i32, fp],
    "uir_pixel_unshuffle": [fp, i32, i32, fp, i32, i32, i32, i32, i32, fp],
    "uir_patch3x3": [fp, i32, fp, i32, i32, i32, i32, i32, fp],
    "uir_patch3x3_bwd": [fp, i32, fp, i32, i32, i32, i32, i32, fp],
    "uir_patch4s2": [fp, i32, fp, i32, i32, i32, i32, i32, fp],
    "uir_patch4s2_bwd": [fp, i32, fp, i32, i32, i32, i32, i32, fp],
    "uir_gather_sum": [fp, fp, fp, i64, i32, fp],
    "uir_copy_cols": [fp, i32, i32, fp, i32, i32, i64, i32, i32, fp],
    "uir_copy_flat": [fp, fp, i64, fp],
    "uir_colsum_partials": [fp, i32, i32, i64, i32, fp, i32, fp],
    "uir_nchw3_to_hwc4": [fp, fp, i32, i32, i32, fp],
    "uir_im2col3x3_c3": [fp, fp, i32, i32, i32, fp],
    "uir_hwc4_to_nchw3": [fp, fp, i32, i32, i32, fp],
    "uir_loss_partials": [fp, fp, fp, i32, i32, i32, fp],
    "uir_loss_finalize": [fp, fp, i32, i32, i32, fp],
    "uir_loss_bwd": [fp, fp, fp, fp, fp, i32, i32, i32, fp],
    "uir_gradloss_partials": [fp, fp, fp, i32, i32, i32, i32, fp],
    "uir_gradloss_bwd": [fp, fp, fp, fp, i32, i32, i32, fp],
    "uir_ffl_diff": [fp, fp, fp, i32, i32, fp],
    "uir_ffl_max": [fp, fp, fp, fp, i32, i32, i32, fp],
    "uir_ffl_partials": [fp, fp, fp, fp, i32, i32, i32, fp],
    "uir_ffl_spectrum_grad": [fp, fp, fp, fp, fp, C.c_float, fp, fp, i32, i32, fp],
    "uir_ffl_add_grad": [fp, fp, i32, i32, i32, fp],
    "uir_msssim_level": [C.POINTER(MsSsimArgs), fp, fp],
    "uir_msssim_combine": [fp, fp, fp, i32, i32, C.POINTER(MsSsimWeights), fp],
    "uir_msssim_level_bwd": [C.POINTER(MsSsimArgs), fp, fp, fp, fp, fp, i32, fp],
    "uir_avgpool2": [fp, fp, i32, i32, i32, fp],
    "uir_avgpool2_bwd": [fp, fp, i32, i32, i32, i32, fp],
    "uir_axpy_dev": [fp, fp, C.c_float, fp, C.c_long, fp],
    "uir_vgg_prep": [fp, fp, i32, i32, i32, i32, fp],
    "uir_vgg_prep_bwd": [fp, fp, fp, C.c_float, i32, i32, i32, i32, fp],
    "uir_maxpool2_hwc": [fp, fp, i32, i32, i32, i32, fp],
    "uir_maxpool2_hwc_bwd": [fp, fp, fp, i32, i32, i32, i32, i32, fp],
    "uir_l1_feat_partials": [fp, fp, fp, i32, C.c_long, fp],
    "uir_l1_feat_bwd": [fp, fp, fp, C.c_float, C.c_long, i32, fp],
    "uir_tv_partials": [fp, fp, i32, i32, i32, i32, fp],
    "uir_tv_bwd": [fp, fp, fp, C.c_float, C.c_float, i32, i32, i32, fp],
    "uir_relu_inplace": [fp, C.c_long, fp],
    "uir_sumsq_partials": [fp, i64, fp, i32, fp],
    "uir_clip_adam_step": [C.POINTER(AdamArgs), fp],
    "uir_adam_set_hyper": [fp, f32, f32, f32, f32, fp],
    "uir_resize_ksize": [i32, i32],                      # host-only helpers: return ksize (or -1), not a status
    "uir_resize_coeffs": [i32, i32, fp, fp, i64],
    "uir_resize_norm_u8": [C.POINTER(ResizeArgs), fp],
    "uir_ssim_nblocks": [i32, i32, i32],                 # host helper: returns a count (c_long), not a status
    "uir_ssim_partials": [C.POINTER(SsimArgs), fp],
    "uir_frfn_act_fwd": [fp, i32, fp, i32, i64, i32, fp],
    "uir_frfn_act_bwd": [fp, i32, fp, i32, fp, i32, i64, i32, fp, i32, fp],
    "uir_leaky_bwd": [fp, i32, fp, i32, fp, i32, i64, i32, f32, fp],
    "uir_frfn_gate_fwd": [C.POINTER(FrfnGateArgs), fp],
    "uir_frfn_gate_bwd": [C.POINTER(FrfnGateArgs), fp],
    "uir_layernorm_fwd": [fp, i32, fp, fp, fp, i32, fp, i64, i32, f32, fp],
    "uir_cmul_table": [fp, fp, fp, fp, i64, i64, i32, fp],
    "uir_transpose_rows": [fp, fp, i32, i32, i32, i32, i32, f32, fp],
    "uir_scale_rows": [fp, i32, fp, i32, fp, i64, i64, i32, i32, fp],
    "uir_add_scaled_rows": [fp, i32, fp, i32, fp, i32, fp, i64, i64, i32, fp],
    "uir_radix2_combine": [fp, fp, fp, fp, fp, fp, i64, i32, i64, i32, f32, fp],
    "uir_winattn_fwd": [C.POINTER(WinAttnArgs), fp],
    "uir_winattn_bwd": [C.POINTER(WinAttnArgs), fp],
    "uir_winattn_wgrad": [fp, fp, fp, fp],
    "uir_uiqm_work_doubles": [i32, i32, i32],            # host helper: returns a count (c_long) or -1
    "uir_uiqm": [C.POINTER(UiqmArgs), fp],
    "uir_stream_create_cu_masked": [i32, C.POINTER(C.c_void_p)],
    "uir_stream_destroy": [fp],
    "uir_event_create": [C.POINTER(fp)],
    "uir_event_destroy": [fp],
    "uir_event_record": [fp, fp],
    "uir_stream_wait_event": [fp, fp],
    "uir_abi_layout": [C.c_char_p, C.POINTER(C.c_long), i32],   # host helper: returns a count (c_long)
}

# C struct name -> ctypes mirror (every argument struct of include/uir_kernels.h)
STRUCTS = {"UirGemmArgs": GemmArgs, "UirGramArgs": GramArgs, "UirConv3bArgs": Conv3bArgs, "UirConv3bWgradArgs": Conv3bWgradArgs, "UirDwArgs": DwArgs, "UirDwWgradArgs": DwWgradArgs,
           "UirDwBwdArgs": DwBwdArgs, "UirGdfnFwdArgs": GdfnFwdArgs, "UirQkvFwdArgs": QkvFwdArgs,
           "UirAttnvFwdArgs": AttnvFwdArgs, "UirAttnvBwdArgs": AttnvBwdArgs, "UirDwLnBwdArgs": DwLnBwdArgs, "UirLnBwdArgs": LnBwdArgs,
           "UirAttnFwdArgs": AttnFwdArgs, "UirAttnBwdArgs": AttnBwdArgs, "UirDftArgs": DftArgs, "UirDft2Args": Dft2Args,
           "UirAdamArgs": AdamArgs, "UirResizeImage": ResizeImage, "UirResizeArgs": ResizeArgs, "UirSsimArgs": SsimArgs,
           "UirUiqmArgs": UiqmArgs, "UirReduceDesc": ReduceDesc,
           "UirFrfnGateArgs": FrfnGateArgs, "UirWinAttnArgs": WinAttnArgs, "UirMsSsimArgs": MsSsimArgs, "UirMsSsimWeights": MsSsimWeights}


def abi_layout(lib_, cname: str):
    """(sizeof, [offsetof ...]) of a C struct as the library was compiled."""
    buf = (C.c_long * 96)()
    n = lib_.uir_abi_layout(cname.encode(), buf, 96)
    if n < 1:
        raise ImportError(f"libuir_hip.so does not know struct {cname}")
    return buf[0], list(buf[1:n])


def _verify_structs(lib_) -> None:
    """Refuse to run with a ctypes mirror that disagrees with the compiled structs (a silent mismatch would hand the
    kernels garbage pointers)."""
    for cname, cls in STRUCTS.items():
        size, offs = abi_layout(lib_, cname)
        mine = [getattr(cls, f[0]).offset for f in cls._fields_]
        if size != C.sizeof(cls) or offs != mine:
            raise ImportError(f"ctypes mirror of {cname} does not match libuir_hip.so: sizeof {C.sizeof(cls)} vs {size}, "
                              f"offsets {mine} vs {offs}")


def _load() -> C.CDLL:
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: build it with `python -m underwater_image_restoration_amd.build` "
            "(or __graft_entry__.build()).  The HIP library is mandatory; there is no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, argtypes in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is missing
        fn.argtypes = argtypes
        fn.restype = C.c_int
    lib.uir_ssim_nblocks.restype = C.c_long
    lib.uir_abi_layout.restype = C.c_long
    lib.uir_uiqm_work_doubles.restype = C.c_long
    _verify_structs(lib)
    lib.uir_version.restype = C.c_char_p
    lib.uir_version.argtypes = []
    lib.uir_last_launch_error.restype = C.c_char_p
    lib.uir_last_launch_error.argtypes = []
    return lib


lib = _load()
from . import knobs as _knobs  # noqa: E402
_knobs.report()                # prints the non-default knobs of this process, once


class UirError(RuntimeError):
    pass


def check(status: int, what: str) -> None:
    if status != 0:
        kind = {1: "argument check failed (UIR_EINVAL)", 2: "HIP launch error (UIR_ELAUNCH)"}.get(status, str(status))
        if status == 2:
            kind += ": " + lib.uir_last_launch_error().decode()
        raise UirError(f"{what}: {kind}")
