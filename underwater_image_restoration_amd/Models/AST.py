"""``--arch AST`` on MI355X behind the reference's surface (src/Models/AST.py): the same classes, constructor
arguments, child names and registration order - hence the same 274 ``state_dict`` tensors and, under a seed, the same initial
values - with every forward / backward a launch list over the C ABI (include/uir_kernels.h):

    InputProj (:447-466)            conv3x3 3 -> C + bias + LeakyReLU           uir_gemm taps=9 (+ uir_leaky_bwd)
    OutputProj (:470-493)           conv3x3 C -> 3 + bias on the token image    uir_gemm taps=9
    Downsample (:408-424)           Conv2d(k 4, s 2, p 1)                       uir_patch4s2 (16-tap patches in the weight's k order)
                                                                                + uir_gemm against weight.view(Cout, 16 Cin) in place
    Upsample (:428-443)             ConvTranspose2d(k 2, s 2)                   uir_gemm C -> 4 C_out + uir_pixel_shuffle
    LeFF (:294-326)                 = src/model/block.py's LeFF                 model.block.record_leff
    WindowAttention_sparse          q/kv Linears, proj                          uir_gemm
      (:144-219)                    scores + bias (+ shift mask), softmax * w1  uir_winattn_fwd / _bwd (args.shift: the cyclic roll
                                    + relu^2 * w2, . v                          of :598-616 is an address, the mask of :568-589 a
                                                                                comparison of region ids)
    TransformerBlock (:498-623)     norm1 -> attention -> + ; norm2 -> LeFF -> +, timm DropPath per branch (uir_add_scaled_rows)
    BasicASTLayer (:629-677), AST (:680-921)

One launch plan per TransformerBlock / sampler / projection (``plan_modules``); tensors travel between them through autograd.
Built for the configuration ``init_model("AST")`` constructs (the reference's defaults); other arguments raise.
There is no CPU / eager fallback.
"""
from __future__ import annotations

import ctypes as C
import math

import numpy as np
import torch
import torch.nn as nn

from .. import _lib as L
from ..engine import _ld, _ptr
from ..model.block import (LeFF, TokenModule, TokenPlan, _SimplePlan, _colsum_wide, record_conv3x3, record_layernorm,
                           record_leff, record_window_attention)
from ..model.model import _ResidualImage

lib = L.lib


# ------------------------------------------------------------------------------------------------ launch-list recorders
def _idx(rec, a: np.ndarray) -> torch.Tensor:
    t = torch.from_numpy(np.ascontiguousarray(a.astype(np.int32))).to(rec.dev)
    rec.keep.append(t)
    return t


def record_ast_downsample(rec, pfx: str, x: torch.Tensor, Cin: int, Cout: int, dims):
    """Downsample.forward (AST.py:417-424): [B*H*W, Cin] -> [B*H*W/4, Cout].  Conv2d(k 4, s 2, p 1) as ONE dense contraction: the
    16-tap patches of every output pixel in the weight's own k order (uir_patch4s2) against weight.view(Cout, Cin * 16) - the weight
    and its gradient are used in place.  (Until round 3 this was a 3 x 3 kernel over the PixelUnshuffle(2) image: 36 taps per
    input channel of which 20 are structural zeros, 2.25 x the arithmetic.)"""
    B, H, W = dims
    H2, W2 = H // 2, W // 2
    M2, K = B * H2 * W2, 16 * Cin
    Wn, bn = rec.P(pfx + "conv.0.weight"), rec.P(pfx + "conv.0.bias")
    P, y = rec.alloc(M2, K), rec.alloc(M2, Cout)
    rec._hold(x, Wn, bn)
    rec._op(pfx + "patches", lib.uir_patch4s2, _ptr(x), _ld(x), _ptr(P), _ld(P), B, H2, W2, Cin, nbytes=4 * (x.numel() + P.numel()))
    rec.gemm(pfx + "conv.0", P, Wn, y, N=Cout, K=K, w_sn=K, w_sk=1, bias=bn)

    def backward(dy):
        dP, dx = rec.alloc(M2, K), rec.alloc(B * H * W, Cin)
        rec._hold(dy)
        with rec.bulk():
            rec.gram(pfx + "conv.0.dW", dy, P, rec.Gr(pfx + "conv.0.weight"), N=Cout, K=K, o_sn=K, o_sk=1)
            _colsum_wide(rec, pfx + "conv.0.db", dy, Cout, rec.Gr(pfx + "conv.0.bias"))
            rec.flush_reductions(pfx + "conv.0.dW")
        rec.gemm(pfx + "conv.0.dX", dy, Wn, dP, N=K, K=Cout, w_sn=1, w_sk=K)
        rec._op(pfx + "patches.bwd", lib.uir_patch4s2_bwd, _ptr(dP), _ld(dP), _ptr(dx), _ld(dx), B, H2, W2, Cin,
                nbytes=4 * (dP.numel() + dx.numel()))
        return dx
    return y, backward


def record_ast_upsample(rec, pfx: str, x: torch.Tensor, Cin: int, Cout: int, dims):
    """Upsample.forward (AST.py:437-443): ConvTranspose2d(k 2, s 2) = one GEMM to the four output phases of every pixel
    (weight [Cin][Cout][2][2] read in place: column n = cout*4 + dy*2 + dx) + PixelShuffle(2): [B*H*W, Cin] -> [4*B*H*W, Cout]."""
    B, H, W = dims
    M, N4 = B * H * W, 4 * Cout
    Wt, bn = rec.P(pfx + "deconv.0.weight"), rec.P(pfx + "deconv.0.bias")
    irep, isum = _idx(rec, np.arange(N4) // 4), _idx(rec, np.arange(N4))
    bx, t, y = rec.alloc(N4), rec.alloc(M, N4), rec.alloc(4 * M, Cout)
    rec._hold(bn)
    rec._op(pfx + "b.layout", lib.uir_gather_sum, _ptr(bn), _ptr(irep), _ptr(bx), N4, 1, nbytes=8 * N4)
    rec.gemm(pfx + "deconv.0", x, Wt, t, N=N4, K=Cin, w_sn=1, w_sk=N4, bias=bx)
    rec._op(pfx + "shuffle", lib.uir_pixel_shuffle, _ptr(t), _ld(t), _ptr(y), _ld(y), 0, B, H, W, Cout, nbytes=8 * t.numel())

    def backward(dy):
        dt, dbx, dx = rec.alloc(M, N4), rec.alloc(N4), rec.alloc(M, Cin)
        rec._hold(dy)
        rec._op(pfx + "shuffle.bwd", lib.uir_pixel_unshuffle, _ptr(dy), _ld(dy), 0, _ptr(dt), _ld(dt), B, H, W, Cout, nbytes=8 * dt.numel())
        with rec.bulk():
            rec.gram(pfx + "deconv.0.dW", dt, x, rec.Gr(pfx + "deconv.0.weight"), N=N4, K=Cin, o_sn=1, o_sk=N4)
            _colsum_wide(rec, pfx + "deconv.0.db4", dt, N4, dbx)
            rec.flush_reductions(pfx + "deconv.0.db4")
            gb = rec.Gr(pfx + "deconv.0.bias")
            rec._op(pfx + "db.layout", lib.uir_gather_sum, _ptr(dbx), _ptr(isum), _ptr(gb), Cout, 4, nbytes=4 * (N4 + Cout))
        rec.gemm(pfx + "deconv.0.dX", dt, Wt, dx, N=Cin, K=N4, w_sn=N4, w_sk=1)
        return dx
    return y, backward


def record_ast_block(rec, x: torch.Tensor, dim: int, heads: int, shift: int, att: bool, dims):
    """TransformerBlock.forward (AST.py:552-623), mask None.  rec.dp_scale [2, B]: the per-sample DropPath factors of the
    attention and the LeFF branch (1 in eval mode / for drop_path 0)."""
    B, H, W = dims
    M, Ltok, C_ = B * H * W, H * W, dim
    rec.dp_scale = rec.alloc(2, B)
    rec.dp_scale.fill_(1.0)
    s_att, s_mlp = rec.dp_scale[0], rec.dp_scale[1]

    def scaled(name, src, dst, s, acc):
        rec._hold(src, dst)
        rec._op(name, lib.uir_scale_rows, _ptr(src), _ld(src), _ptr(dst), _ld(dst), _ptr(s), Ltok, M, C_, acc, nbytes=(12 if acc else 8) * M * C_)

    y = x
    if att:
        xn, n1_bw = record_layernorm(rec, "norm1.", x, C_)
        # x + drop_path(attn(...)), y + drop_path(mlp(...)) (AST.py:617,620): shortcut and per-sample factor are the epilogue of the
        # branch's last Linear (proj / linear2)
        y, a_bw = record_window_attention(rec, "attn.", xn, None, dims, C_, heads, qname="qkv.to_q", kvname="qkv.to_kv", shift=shift,
                                          residual=(x, s_att, Ltok))
    x2, n2_bw = record_layernorm(rec, "norm2.", y, C_)
    out, m_bw = record_leff(rec, "mlp.", x2, C_, 4 * C_, dims, residual=(y, s_mlp, Ltok))

    def backward(dout):
        dm = rec.alloc(M, C_)
        scaled("dmlp=dp*dout", dout, dm, s_mlp, 0)
        dy = n2_bw(m_bw(dm), dout)                          # y feeds norm2 -> LeFF and the residual
        if not att:
            return dy
        da = rec.alloc(M, C_)
        scaled("dattn=dp*dy", dy, da, s_att, 0)
        dxn, _ = a_bw(da)
        return n1_bw(dxn, dy)                               # x feeds norm1 -> attention and the shortcut
    return out, backward


class _BlockPlan(TokenPlan):
    def __init__(self, fp, dim, heads, shift, att, B, H, W, want_backward=True):
        self.heads, self.shift, self.att = heads, shift, att
        super().__init__(fp, dim, B, H, W, want_backward)

    def _record(self, x):
        return record_ast_block(self, x, self.dim, self.heads, self.shift, self.att, (self.B, self.H, self.W))


# ------------------------------------------------------------------------------------------------ the reference's classes
class LinearProjection(nn.Module):
    """Parameter container of AST.py:42-50 (``to_q``, ``to_kv``)."""

    def __init__(self, dim, heads=8, dim_head=64, dropout=0., bias=True):
        super().__init__()
        inner_dim = dim_head * heads
        self.heads = heads
        self.to_q = nn.Linear(dim, inner_dim, bias=bias)
        self.to_kv = nn.Linear(dim, inner_dim * 2, bias=bias)
        self.dim = dim
        self.inner_dim = inner_dim


class WindowAttention_sparse(nn.Module):
    """Parameter container with the names and registration order of AST.py:144-185; its arithmetic runs inside the owning
    TransformerBlock's launch plan."""

    def __init__(self, dim, win_size, num_heads, token_projection='linear', qkv_bias=True, qk_scale=None, attn_drop=0., proj_drop=0.):
        super().__init__()
        if tuple(win_size) != (8, 8) or token_projection != 'linear' or not qkv_bias or qk_scale is not None or attn_drop or proj_drop:
            raise NotImplementedError("the HIP window attention is built for 8 x 8 windows, linear projection with bias, no dropout")
        self.dim, self.win_size, self.num_heads = dim, tuple(win_size), num_heads
        self.scale = (dim // num_heads) ** -0.5
        ws = 8
        self.relative_position_bias_table = nn.Parameter(torch.zeros((2 * ws - 1) * (2 * ws - 1), num_heads))
        c = torch.stack(torch.meshgrid([torch.arange(ws), torch.arange(ws)], indexing="ij")).flatten(1)
        rel = (c[:, :, None] - c[:, None, :]).permute(1, 2, 0).contiguous()
        rel[:, :, 0] += ws - 1
        rel[:, :, 1] += ws - 1
        rel[:, :, 0] *= 2 * ws - 1
        self.register_buffer("relative_position_index", rel.sum(-1))
        nn.init.trunc_normal_(self.relative_position_bias_table, std=.02)
        self.qkv = LinearProjection(dim, num_heads, dim // num_heads, bias=qkv_bias)
        self.token_projection = token_projection
        self.proj = nn.Linear(dim, dim)
        self.w = nn.Parameter(torch.ones(2))


class TransformerBlock(TokenModule):
    """Drop-in for AST.py's TransformerBlock (:498-623) as BasicASTLayer builds it: win_size 8, shift 0 / 4, LeFF, LayerNorm,
    ``att`` False (encoders: LeFF only) or True with ``sparseAtt`` (bottleneck, decoders)."""

    def __init__(self, dim, input_resolution, num_heads, win_size=8, shift_size=0, mlp_ratio=4., qkv_bias=True, qk_scale=None, drop=0.,
                 attn_drop=0., drop_path=0., act_layer=nn.GELU, norm_layer=nn.LayerNorm, token_projection='linear', token_mlp='leff',
                 att=True, sparseAtt=False):
        super().__init__()
        if token_mlp not in ('ffn', 'mlp', 'leff', 'frfn'):
            raise Exception("FFN error!")                                                     # AST.py:543
        if token_mlp != 'leff' or (att and not sparseAtt) or win_size != 8 or mlp_ratio != 4 or drop or attn_drop \
                or act_layer is not nn.GELU or norm_layer is not nn.LayerNorm:
            raise NotImplementedError("the HIP TransformerBlock is built for AST()'s defaults: LeFF, sparse window attention, win_size 8, "
                                      "mlp_ratio 4, no dropout")
        if dim not in (32, 64, 128, 256, 512) or (att and dim // num_heads not in (8, 16, 32, 64, 128)):
            raise NotImplementedError("dim in {32, 64, 128, 256, 512} (the LayerNorm kernels), head dims 8 ... 128")
        self.att, self.sparseAtt = att, sparseAtt
        self.dim, self.input_resolution, self.num_heads = dim, input_resolution, num_heads
        self.win_size, self.shift_size, self.mlp_ratio, self.token_mlp = win_size, shift_size, mlp_ratio, token_mlp
        if min(self.input_resolution) <= self.win_size:                                      # :515-517
            self.shift_size = 0
            self.win_size = min(self.input_resolution)
            if self.win_size != 8:
                raise NotImplementedError("8 x 8 windows")
        assert 0 <= self.shift_size < self.win_size, "shift_size must in 0-win_size"
        if self.att:                                                                          # registration order :520-544
            self.norm1 = norm_layer(dim)
            self.attn = WindowAttention_sparse(dim, win_size=(self.win_size, self.win_size), num_heads=num_heads, qkv_bias=qkv_bias,
                                               qk_scale=qk_scale, attn_drop=attn_drop, proj_drop=drop, token_projection=token_projection)
        self.drop_prob = float(drop_path)
        self.drop_path = nn.Identity()                  # the stochastic-depth factors are applied inside the launch plan
        self.norm2 = norm_layer(dim)
        self.mlp = LeFF(dim, int(dim * mlp_ratio), act_layer=act_layer, drop=drop)
        self._init_plumbing()

    def _make_plan(self, fp, B, H, W, want_backward):
        if H % 8 or W % 8:
            raise ValueError("token image sides must be multiples of the 8 x 8 window")
        return _BlockPlan(fp, self.dim, self.num_heads, self.shift_size, self.att, B, H, W, want_backward)

    def _before_forward(self, plan) -> None:
        """timm's drop_path (imported at AST.py:4): in training mode with p > 0 each residual branch gets an independent
        per-sample factor Bernoulli(1 - p) / (1 - p), drawn in the reference's order (attention, then LeFF: :617,620)."""
        if self.training and self.drop_prob > 0.0:
            keep = 1.0 - self.drop_prob
            for r in ((0, 1) if self.att else (1,)):
                plan.dp_scale[r].copy_(torch.empty(plan.B, device=plan.dp_scale.device).bernoulli_(keep).div_(keep))
        else:
            plan.dp_scale.fill_(1.0)

    def forward(self, x, mask=None):
        if mask is not None:
            raise NotImplementedError("mask=None (the trainer never passes one, src/ModelTrainer.py:79)")
        return super().forward(x)


class BasicASTLayer(nn.Module):
    """AST.py:629-677: ``depth`` TransformerBlocks, every second one shifted by win_size // 2."""

    def __init__(self, dim, output_dim, input_resolution, depth, num_heads, win_size, mlp_ratio=4., qkv_bias=True, qk_scale=None, drop=0.,
                 attn_drop=0., drop_path=0., norm_layer=nn.LayerNorm, use_checkpoint=False, token_projection='linear', token_mlp='ffn',
                 shift_flag=True, att=False, sparseAtt=False):
        super().__init__()
        if use_checkpoint:
            raise NotImplementedError("use_checkpoint=False (a launch plan keeps its activations)")
        self.att, self.sparseAtt, self.dim, self.input_resolution, self.depth, self.use_checkpoint = att, sparseAtt, dim, input_resolution, depth, False
        self.blocks = nn.ModuleList([
            TransformerBlock(dim=dim, input_resolution=input_resolution, num_heads=num_heads, win_size=win_size,
                             shift_size=0 if (i % 2 == 0 or not shift_flag) else win_size // 2, mlp_ratio=mlp_ratio, qkv_bias=qkv_bias,
                             qk_scale=qk_scale, drop=drop, attn_drop=attn_drop,
                             drop_path=drop_path[i] if isinstance(drop_path, list) else drop_path, norm_layer=norm_layer,
                             token_projection=token_projection, token_mlp=token_mlp, att=att, sparseAtt=sparseAtt)
            for i in range(depth)])

    def forward(self, x, mask=None):
        for blk in self.blocks:
            x = blk(x, mask)
        return x


class Downsample(TokenModule):
    """AST.py:408-424: (B, L, C_in) -> (B, L/4, C_out)."""

    def __init__(self, in_channel, out_channel):
        super().__init__()
        if in_channel % 4 or out_channel % 4:
            raise NotImplementedError("channel counts must be multiples of 4")
        self.conv = nn.Sequential(nn.Conv2d(in_channel, out_channel, kernel_size=4, stride=2, padding=1))
        self.in_channel, self.out_channel = in_channel, out_channel
        self.dim = in_channel
        self._init_plumbing()

    def _make_plan(self, fp, B, H, W, want_backward):
        return _SimplePlan(fp, self.dim, B, H, W, want_backward,
                           lambda rec, x: record_ast_downsample(rec, "", x, self.in_channel, self.out_channel, (B, H, W)))


class Upsample(TokenModule):
    """AST.py:428-443: (B, L, C_in) -> (B, 4L, C_out)."""

    def __init__(self, in_channel, out_channel):
        super().__init__()
        if in_channel % 4 or out_channel % 4:
            raise NotImplementedError("channel counts must be multiples of 4")
        self.deconv = nn.Sequential(nn.ConvTranspose2d(in_channel, out_channel, kernel_size=2, stride=2))
        self.in_channel, self.out_channel = in_channel, out_channel
        self.dim = in_channel
        self._init_plumbing()

    def _make_plan(self, fp, B, H, W, want_backward):
        return _SimplePlan(fp, self.dim, B, H, W, want_backward,
                           lambda rec, x: record_ast_upsample(rec, "", x, self.in_channel, self.out_channel, (B, H, W)))


def _record_input_proj(rec, img_nchw, C_, dims):
    B, H, W = dims
    M = B * H * W
    x4 = rec.alloc(M, 4)
    rec._hold(img_nchw)
    rec._op("to_hwc", lib.uir_nchw3_to_hwc4, _ptr(img_nchw), _ptr(x4), B, H, W, nbytes=28 * M)
    y, bw = record_conv3x3(rec, "proj.0", x4, rec.P("proj.0.weight"), rec.P("proj.0.bias"), 3, C_, dims, Kp=4, k_grp=(3, 4),
                           act=L.ACT_LEAKY, slope=0.01, need_dx=False)

    def backward(dy):
        bw(dy, rec.Gr("proj.0.weight"), rec.Gr("proj.0.bias"))
        return None
    return y, backward


def _record_output_proj(rec, x, C_, dims):
    B, H, W = dims
    M = B * H * W
    c, bw = record_conv3x3(rec, "proj.0", x, rec.P("proj.0.weight"), rec.P("proj.0.bias"), C_, 3, dims, Np=4, n_grp=(3, 4))
    out = rec.alloc(B, 3, H, W)
    rec._op("to_nchw", lib.uir_hwc4_to_nchw3, _ptr(c), _ptr(out), B, H, W, nbytes=28 * M)

    def backward(dout_nchw):
        dc = rec.alloc(M, 4)
        rec._hold(dout_nchw)
        rec._op("to_hwc.bwd", lib.uir_nchw3_to_hwc4, _ptr(dout_nchw), _ptr(dc), B, H, W, nbytes=28 * M)
        return bw(dc, rec.Gr("proj.0.weight"), rec.Gr("proj.0.bias"))
    return out, backward


class InputProj(TokenModule):
    """AST.py:447-466 (norm_layer None): ``forward(x: (B, 3, H, W)) -> (B, H*W, C)``."""

    def __init__(self, in_channel=3, out_channel=64, kernel_size=3, stride=1, norm_layer=None, act_layer=nn.LeakyReLU):
        super().__init__()
        if (in_channel, kernel_size, stride, norm_layer) != (3, 3, 1, None) or act_layer is not nn.LeakyReLU or out_channel % 4:
            raise NotImplementedError("the HIP InputProj is built for 3 input channels, kernel 3, stride 1, LeakyReLU, no norm")
        self.proj = nn.Sequential(nn.Conv2d(in_channel, out_channel, kernel_size=3, stride=stride, padding=kernel_size // 2),
                                  act_layer(inplace=True))
        self.norm = None
        self.in_channel, self.out_channel = in_channel, out_channel
        self.dim = out_channel
        self._init_plumbing()

    def _grid_of(self, x):
        return x.shape[0], x.shape[2], x.shape[3]

    def _out_shape(self, x, plan):
        return (x.shape[0], x.shape[2] * x.shape[3], self.out_channel)

    def _make_plan(self, fp, B, H, W, want_backward):
        return _SimplePlan(fp, self.out_channel, B, H, W, want_backward,
                           lambda rec, x: _record_input_proj(rec, x, self.out_channel, (B, H, W)), in_shape=(B, 3, H, W))

    def forward(self, x):
        if x.dim() != 4 or x.shape[1] != 3:
            raise ValueError(f"expected (B, 3, H, W), got {tuple(x.shape)}")
        if x.device.type != "cuda":
            raise RuntimeError("InputProj(HIP) runs on an AMD GPU ('cuda' device in PyTorch-ROCm); there is no CPU fallback")
        x = x.contiguous().float()
        self._ensure_flat(x.device)
        from ..model.block import _TokenFunction
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            return _TokenFunction.apply(x, self, *self.parameters())
        plan = self._plan_for(x, want_backward=False)
        self._serial += 1
        y = plan.forward(x)
        return (y if self._alias_io else y.clone()).view(self._out_shape(x, plan))


class OutputProj(TokenModule):
    """AST.py:470-493 (no act / norm): ``forward(x: (B, L, C)) -> (B, 3, H, W)``."""

    def __init__(self, in_channel=64, out_channel=3, kernel_size=3, stride=1, norm_layer=None, act_layer=None):
        super().__init__()
        if (out_channel, kernel_size, stride, norm_layer, act_layer) != (3, 3, 1, None, None) or in_channel % 4:
            raise NotImplementedError("the HIP OutputProj is built for 3 output channels, kernel 3, stride 1, no norm / act")
        self.proj = nn.Sequential(nn.Conv2d(in_channel, out_channel, kernel_size=3, stride=stride, padding=kernel_size // 2))
        self.norm = None
        self.in_channel, self.out_channel = in_channel, out_channel
        self.dim = in_channel
        self._init_plumbing()

    def _out_shape(self, x, plan):
        hh = int(math.sqrt(x.shape[1]))
        return (x.shape[0], 3, hh, hh)

    def _make_plan(self, fp, B, H, W, want_backward):
        return _SimplePlan(fp, self.dim, B, H, W, want_backward, lambda rec, x: _record_output_proj(rec, x, self.dim, (B, H, W)))


class AST(nn.Module):
    """``--arch AST`` (AST.py:680-921) with the reference's defaults: embed_dim 32, nine BasicASTLayers of depth 2 (encoders LeFF
    only, bottleneck and decoders with sparse window attention, every second block shifted), stochastic depth 0 ... 0.1."""

    # device memory the 28 launch plans keep per input pixel (measured, tests/test_ast.py holds it): ModuleTrainStep sizes micro-batches with it
    # (tools/measure_module_plans.py, round 3 with LeFF's activations fused away and the residual sums as GEMM epilogues: 2.8 / 9.6 /
    # 17.3 GiB at 1 x 128^2 / 2 x 256^2 / 1 x 512^2, 61.2 GiB at 4 x 512^2 = 60 KB per pixel + 2.9 GB of per-plan scratch)
    PLAN_BYTES_PER_PIXEL_TRAIN, PLAN_BYTES_FIXED_TRAIN = 61_000, 2_900_000_000
    PLAN_BYTES_PER_PIXEL_EVAL, PLAN_BYTES_FIXED_EVAL = 30_800, 2_000_000_000

    def __init__(self, img_size=256, in_chans=3, dd_in=3, embed_dim=32, depths=[2, 2, 2, 2, 2, 2, 2, 2, 2],
                 num_heads=[1, 2, 4, 8, 16, 16, 8, 4, 2], win_size=8, mlp_ratio=4., qkv_bias=True, qk_scale=None, drop_rate=0.,
                 attn_drop_rate=0., drop_path_rate=0.1, norm_layer=nn.LayerNorm, patch_norm=True, use_checkpoint=False,
                 token_projection='linear', token_mlp='leff', dowsample=Downsample, upsample=Upsample, shift_flag=True, **kwargs):
        super().__init__()
        if (in_chans, dd_in) != (3, 3) or drop_rate or attn_drop_rate or dowsample is not Downsample or upsample is not Upsample:
            raise NotImplementedError("the HIP AST is built for RGB in / out (dd_in = 3), no dropout, the module's own samplers")
        self.num_enc_layers = self.num_dec_layers = len(depths) // 2
        self.embed_dim, self.patch_norm, self.mlp_ratio, self.token_projection, self.mlp = embed_dim, patch_norm, mlp_ratio, token_projection, token_mlp
        self.win_size, self.reso, self.dd_in = win_size, img_size, dd_in
        self.pos_drop = nn.Dropout(p=drop_rate)
        enc_dpr = [x.item() for x in torch.linspace(0, drop_path_rate, sum(depths[:self.num_enc_layers]))]     # :705-707
        conv_dpr = [drop_path_rate] * depths[4]
        dec_dpr = enc_dpr[::-1]
        E_, r = embed_dim, img_size

        def layer(i, mult, res, dpr, att):
            return BasicASTLayer(dim=E_ * mult, output_dim=E_ * mult, input_resolution=(res, res), depth=depths[i], num_heads=num_heads[i],
                                 win_size=win_size, mlp_ratio=mlp_ratio, qkv_bias=qkv_bias, qk_scale=qk_scale, drop=drop_rate,
                                 attn_drop=attn_drop_rate, drop_path=dpr, norm_layer=norm_layer, use_checkpoint=use_checkpoint,
                                 token_projection=token_projection, token_mlp=token_mlp, shift_flag=shift_flag, att=att, sparseAtt=att)
        # registration order == reference (:712-861)
        self.input_proj = InputProj(in_channel=dd_in, out_channel=E_, kernel_size=3, stride=1, act_layer=nn.LeakyReLU)
        self.output_proj = OutputProj(in_channel=2 * E_, out_channel=in_chans, kernel_size=3, stride=1)
        self.encoderlayer_0 = layer(0, 1, r, enc_dpr[sum(depths[:0]):sum(depths[:1])], False)
        self.dowsample_0 = dowsample(E_, E_ * 2)
        self.encoderlayer_1 = layer(1, 2, r // 2, enc_dpr[sum(depths[:1]):sum(depths[:2])], False)
        self.dowsample_1 = dowsample(E_ * 2, E_ * 4)
        self.encoderlayer_2 = layer(2, 4, r // 4, enc_dpr[sum(depths[:2]):sum(depths[:3])], False)
        self.dowsample_2 = dowsample(E_ * 4, E_ * 8)
        self.encoderlayer_3 = layer(3, 8, r // 8, enc_dpr[sum(depths[:3]):sum(depths[:4])], False)
        self.dowsample_3 = dowsample(E_ * 8, E_ * 16)
        self.conv = layer(4, 16, r // 16, conv_dpr, True)
        self.upsample_0 = upsample(E_ * 16, E_ * 8)
        self.decoderlayer_0 = layer(5, 16, r // 8, dec_dpr[:depths[5]], True)
        self.upsample_1 = upsample(E_ * 16, E_ * 4)
        self.decoderlayer_1 = layer(6, 8, r // 4, dec_dpr[sum(depths[5:6]):sum(depths[5:7])], True)
        self.upsample_2 = upsample(E_ * 8, E_ * 2)
        self.decoderlayer_2 = layer(7, 4, r // 2, dec_dpr[sum(depths[5:7]):sum(depths[5:8])], True)
        self.upsample_3 = upsample(E_ * 4, E_)
        self.decoderlayer_3 = layer(8, 2, r, dec_dpr[sum(depths[5:8]):sum(depths[5:9])], True)
        self.apply(self._init_weights)

    def _init_weights(self, m):                                                                 # :865-873
        if isinstance(m, nn.Linear):
            nn.init.trunc_normal_(m.weight, std=.02)       # timm's trunc_normal_ is PyTorch's algorithm
            if m.bias is not None:
                nn.init.constant_(m.bias, 0)
        elif isinstance(m, nn.LayerNorm):
            nn.init.constant_(m.bias, 0)
            nn.init.constant_(m.weight, 1.0)

    @torch.jit.ignore
    def no_weight_decay(self):
        return {'absolute_pos_embed'}

    @torch.jit.ignore
    def no_weight_decay_keywords(self):
        return {'relative_position_bias_table'}

    def forward(self, x, mask=None):
        if mask is not None:
            raise NotImplementedError("mask=None (the trainer never passes one, src/ModelTrainer.py:79)")
        if x.dim() != 4 or x.shape[1] != 3 or x.shape[2] != x.shape[3] or x.shape[2] % 128:
            raise ValueError(f"expected (B, 3, S, S) with S a multiple of 128 (four 2x down-samplings to at least one 8 x 8 window), got {tuple(x.shape)}")
        y = self.pos_drop(self.input_proj(x))
        conv0 = self.encoderlayer_0(y)
        conv1 = self.encoderlayer_1(self.dowsample_0(conv0))
        conv2 = self.encoderlayer_2(self.dowsample_1(conv1))
        conv3 = self.encoderlayer_3(self.dowsample_2(conv2))
        conv4 = self.conv(self.dowsample_3(conv3))
        deconv0 = self.decoderlayer_0(torch.cat([self.upsample_0(conv4), conv3], -1))
        deconv1 = self.decoderlayer_1(torch.cat([self.upsample_1(deconv0), conv2], -1))
        deconv2 = self.decoderlayer_2(torch.cat([self.upsample_2(deconv1), conv1], -1))
        deconv3 = self.decoderlayer_3(torch.cat([self.upsample_3(deconv2), conv0], -1))
        return _ResidualImage.apply(self.output_proj(deconv3), x.contiguous().float())         # x + y (:921)

    def plan_modules(self):
        """The launch-plan modules, in registration order (each owns one flat parameter / gradient buffer)."""
        out = []
        for m in self.children():
            if isinstance(m, BasicASTLayer):
                out += list(m.blocks)
            elif isinstance(m, TokenModule):
                out.append(m)
        return out
