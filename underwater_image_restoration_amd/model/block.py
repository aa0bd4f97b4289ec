"""``FRFN`` of the NewBigFRFNModel family on MI355X behind the reference's surface
(src/model/block.py:249-282): same constructor, same seven ``state_dict`` tensors in the same registration order
(``linear1.0.{weight,bias}``, ``dwconv.0.{weight,bias}``, ``linear2.0.{weight,bias}``, ``partial_conv3.weight``), same
``forward(x: (B, L, C)) -> (B, L, C)`` with L = H*W a square - but forward and backward are launch lists over the C ABI:

    partial 3x3 conv on the first C/4 channels (:267-270)      uir_gemm taps=9 on a column slice (+ uir_copy_cols for the rest)
    Linear C -> 2*hidden + bias, GELU (:252,273)                uir_gemm (f32 MFMA, bias epilogue) + uir_frfn_act_fwd
    depthwise 3x3 (+ bias, GELU) on the x_1 half (:253,276-278) uir_dwconv, bias + GELU folded into the gate pass
    gate x_1 * x_2 (:279)                                       uir_frfn_gate_fwd
    Linear hidden -> C + bias (:254,281)                        uir_gemm
    backward                                                    uir_gemm (transposed weights), uir_frfn_gate_bwd / _act_bwd,
                                                                uir_dwconv_bwd_fused, uir_gram + uir_colsum_partials on
                                                                the weight-gradient lane, batched slab reductions

The reference's ``(B, L, C)`` token tensor IS the HWC row layout of the kernels: no transposes anywhere (the reference
rearranges to NCHW and back four times per call).  There is no CPU / eager fallback.
"""
from __future__ import annotations

import math
from typing import Dict, Tuple

import torch
import torch.nn as nn

from .. import _lib as L
from ..engine import FlatParams, Recorder, _ptr, _ld
import ctypes as C

lib = L.lib


_DB_BLOCKS = 1024      # workgroups (= slab rows) of the element-wise backward passes that also sum the bias gradient


def _fused_db(cols: int) -> bool:
    return cols <= 1024 and cols % 4 == 0 and 256 % (cols // 4) == 0


def _act(rec, name, pre, t, cols, bwd_dt=None, dpre=None, dbias=None):
    rec._hold(pre, t, bwd_dt, dpre)
    M = pre.shape[0]
    if bwd_dt is None:
        rec._op(name, lib.uir_frfn_act_fwd, _ptr(pre), _ld(pre), _ptr(t), _ld(t), M, cols, nbytes=8 * M * cols)
        return
    slab = rec.alloc(_DB_BLOCKS, cols) if dbias is not None and _fused_db(cols) else None
    rec._op(name, lib.uir_frfn_act_bwd, _ptr(pre), _ld(pre), _ptr(bwd_dt), _ld(bwd_dt), _ptr(dpre), _ld(dpre), M, cols,
            _ptr(slab), _DB_BLOCKS if slab is not None else 0, nbytes=12 * M * cols)
    if dbias is not None:
        with rec.bulk():
            if slab is not None:      # the bias gradient came out of the same pass: only its (batched) slab reduction is left
                rec._reduce(name + ".db", _ptr(slab), _DB_BLOCKS, cols, _ptr(dbias), cols)
            else:
                _colsum_wide(rec, name + ".db", dpre, cols, dbias)


def _gate(rec, name, hid, U, bias, V, G=None, dG=None, dU=None, dV=None, dbias=None):
    """V = None: no gate (LeFF): G = GELU(U + bias)."""
    rec._hold(U, bias, V, G, dG, dU, dV)
    a = L.FrfnGateArgs()
    a.U, a.bias = _ptr(U), _ptr(bias)
    a.V = _ptr(V) if V is not None else None
    a.M, a.hid = U.shape[0], hid
    a.ldu, a.ldv = _ld(U), (_ld(V) if V is not None else 0)
    rec.keep.append(a)
    if dG is None:
        a.G, a.ldg = _ptr(G), _ld(G)
        rec._op(name, lib.uir_frfn_gate_fwd, C.byref(a), nbytes=(12 if V is not None else 8) * a.M * a.hid)
        return
    a.dG, a.ldg = _ptr(dG), _ld(dG)
    a.dU, a.lddu = _ptr(dU), _ld(dU)
    if dV is not None:
        a.dV, a.lddv = _ptr(dV), _ld(dV)
    slab = rec.alloc(_DB_BLOCKS, hid) if dbias is not None and _fused_db(hid) else None
    if slab is not None:
        a.db_slab, a.db_blocks = _ptr(slab), _DB_BLOCKS
    rec._op(name, lib.uir_frfn_gate_bwd, C.byref(a), nbytes=(20 if V is not None else 12) * a.M * a.hid)
    if dbias is not None:
        with rec.bulk():
            if slab is not None:
                rec._reduce(name + ".db", _ptr(slab), _DB_BLOCKS, hid, _ptr(dbias), hid)
            else:
                _colsum_wide(rec, name + ".db", dU, hid, dbias)


def _colsum_wide(rec, name, src, cols, out):
    """Bias gradient over `cols` columns: uir_colsum_partials handles <= 256 columns per launch."""
    for c0 in range(0, cols, 256):
        n = min(256, cols - c0)
        rec.colsum(f"{name}[{c0}]", src, c0, n, out[c0:c0 + n])


def record_frfn(rec: Recorder, pfx: str, x: torch.Tensor, dim: int, hid: int, dims):
    """Record FRFN.forward (block.py:262-282) on token rows x [M, dim] into `rec` (parameters `pfx + name` of rec.fp).
    Returns (y, backward) with backward(dy) -> dx recording the backward launches."""
    B, H, W = dims
    M, C_, c4 = B * H * W, dim, dim // 4
    P, Gr = (lambda n: rec.P(pfx + n)), (lambda n: rec.Gr(pfx + n))
    W1, b1 = P("linear1.0.weight"), P("linear1.0.bias")
    Wdw, bdw = P("dwconv.0.weight"), P("dwconv.0.bias")
    W2, b2 = P("linear2.0.weight"), P("linear2.0.bias")
    Wpc = P("partial_conv3.weight")
    xc, pre, t = rec.alloc(M, C_), rec.alloc(M, 2 * hid), rec.alloc(M, 2 * hid)
    u, g, y = rec.alloc(M, hid), rec.alloc(M, hid), rec.alloc(M, C_)
    n = pfx
    # partial_conv3 on channels [0, C/4), the rest passes through (block.py:267-270)
    rec.copy_cols(n + "pc3.pass", x, c4, xc, c4, C_ - c4)
    rec.gemm(n + "pc3", x[:, :c4], Wpc, xc[:, :c4], N=c4, K=c4, w_sn=c4 * 9, w_sk=9, w_st=1, taps=9, hw=(H, W))
    # Linear + bias + GELU in one pass: the GEMM epilogue stores the pre-activation (for the backward) and GELU of it
    rec.gemm(n + "linear1+gelu", xc, W1, t, N=2 * hid, K=C_, w_sn=C_, w_sk=1, bias=b1, act=L.ACT_GELU, pre_out=pre)
    rec.dwconv(n + "dwconv", t[:, :hid], Wdw, u, C_=hid, dims=dims)
    _gate(rec, n + "dwconv.gelu*x2", hid, u, bdw, t[:, hid:], G=g)
    rec.gemm(n + "linear2", g, W2, y, N=C_, K=hid, w_sn=hid, w_sk=1, bias=b2)

    def backward(dy: torch.Tensor) -> torch.Tensor:
        dx = rec.alloc(M, C_)
        dg, du, dt = rec.alloc(M, hid), rec.alloc(M, hid), rec.alloc(M, 2 * hid)
        dpre, dxc = rec.alloc(M, 2 * hid), rec.alloc(M, C_)
        with rec.bulk():
            rec.gram(n + "linear2.dW", dy, g, Gr("linear2.0.weight"), N=C_, K=hid, o_sn=hid, o_sk=1)
            _colsum_wide(rec, n + "linear2.db", dy, C_, Gr("linear2.0.bias"))
        rec.gemm(n + "linear2.dX", dy, W2, dg, N=hid, K=C_, w_sn=1, w_sk=hid)
        _gate(rec, n + "gate.bwd", hid, u, bdw, t[:, hid:], dG=dg, dU=du, dV=dt[:, hid:], dbias=Gr("dwconv.0.bias"))
        rec.dwconv_bwd(n + "dwconv.bwd", du, t[:, :hid], Wdw, dt[:, :hid], Gr("dwconv.0.weight"), C_=hid, creal=hid, dims=dims)
        _act(rec, n + "linear1.gelu.bwd", pre, None, 2 * hid, bwd_dt=dt, dpre=dpre, dbias=Gr("linear1.0.bias"))
        with rec.bulk():
            rec.gram(n + "linear1.dW", dpre, xc, Gr("linear1.0.weight"), N=2 * hid, K=C_, o_sn=C_, o_sk=1)
        rec.gemm(n + "linear1.dX", dpre, W1, dxc, N=C_, K=2 * hid, w_sn=1, w_sk=C_)
        rec.copy_cols(n + "pc3.pass.bwd", dxc, c4, dx, c4, C_ - c4)
        with rec.bulk():
            rec.gram(n + "pc3.dW", dxc[:, :c4], x[:, :c4], Gr("partial_conv3.weight"), N=c4, K=c4, o_sn=c4 * 9, o_sk=9, o_st=1,
                     taps=9, hw=(H, W))
        rec.gemm(n + "pc3.dX", dxc[:, :c4], Wpc, dx[:, :c4], N=c4, K=c4, w_sn=9, w_sk=c4 * 9, w_st=1, taps=9, flip=1, hw=(H, W))
        return dx
    return y, backward


def record_leff(rec: Recorder, pfx: str, x: torch.Tensor, dim: int, hid: int, dims):
    """Record LeFF.forward (block.py:234-247): Linear C -> hidden + GELU, depthwise 3x3 + bias + GELU, Linear hidden -> C."""
    B, H, W = dims
    M, C_ = B * H * W, dim
    P, Gr = (lambda n: rec.P(pfx + n)), (lambda n: rec.Gr(pfx + n))
    W1, b1 = P("linear1.0.weight"), P("linear1.0.bias")
    Wdw, bdw = P("dwconv.0.weight"), P("dwconv.0.bias")
    W2, b2 = P("linear2.0.weight"), P("linear2.0.bias")
    pre, t, u, g, y = rec.alloc(M, hid), rec.alloc(M, hid), rec.alloc(M, hid), rec.alloc(M, hid), rec.alloc(M, C_)
    n = pfx
    rec.gemm(n + "linear1+gelu", x, W1, t, N=hid, K=C_, w_sn=C_, w_sk=1, bias=b1, act=L.ACT_GELU, pre_out=pre)
    rec.dwconv(n + "dwconv", t, Wdw, u, C_=hid, dims=dims)
    _gate(rec, n + "dwconv.gelu", hid, u, bdw, None, G=g)
    rec.gemm(n + "linear2", g, W2, y, N=C_, K=hid, w_sn=hid, w_sk=1, bias=b2)

    def backward(dy: torch.Tensor) -> torch.Tensor:
        dg, du, dt, dpre, dx = rec.alloc(M, hid), rec.alloc(M, hid), rec.alloc(M, hid), rec.alloc(M, hid), rec.alloc(M, C_)
        with rec.bulk():
            rec.gram(n + "linear2.dW", dy, g, Gr("linear2.0.weight"), N=C_, K=hid, o_sn=hid, o_sk=1)
            _colsum_wide(rec, n + "linear2.db", dy, C_, Gr("linear2.0.bias"))
        rec.gemm(n + "linear2.dX", dy, W2, dg, N=hid, K=C_, w_sn=1, w_sk=hid)
        _gate(rec, n + "dwconv.gelu.bwd", hid, u, bdw, None, dG=dg, dU=du, dbias=Gr("dwconv.0.bias"))
        rec.dwconv_bwd(n + "dwconv.bwd", du, t, Wdw, dt, Gr("dwconv.0.weight"), C_=hid, creal=hid, dims=dims)
        _act(rec, n + "linear1.gelu.bwd", pre, None, hid, bwd_dt=dt, dpre=dpre, dbias=Gr("linear1.0.bias"))
        with rec.bulk():
            rec.gram(n + "linear1.dW", dpre, x, Gr("linear1.0.weight"), N=hid, K=C_, o_sn=C_, o_sk=1)
        rec.gemm(n + "linear1.dX", dpre, W1, dx, N=C_, K=hid, w_sn=1, w_sk=C_)
        return dx
    return y, backward


def record_spatial_refft(rec: Recorder, name: str, x: torch.Tensor, dims, C_: int, scale: float) -> torch.Tensor:
    """y = scale * Re(fft2 over (H, W)) per channel of token-image rows x [B*H*W, C] = scale * (C_H x C_W - S_H x S_W):
    `fft.fftn(x, dim=(-2,-1)).real` of FDFP on NCHW (block.py:538) and, divided by H*W, its `ifftn(...).real` (:549) - the
    operator is real-symmetric, so it is also its own backward pass.  Two uir_dft_axis passes (W, then H: real part only)."""
    B, H, W = dims
    M = B * H * W
    ar, ai, y = rec.alloc(M, C_), rec.alloc(M, C_), rec.alloc(M, C_)
    rec.dft(name + ".W", x, None, ar, ai, O=B * H, N=W, I=1, C_=C_, sign=1.0, scale=1.0)
    rec.dft(name + ".H", ar, ai, y, None, O=B, N=H, I=W, C_=C_, sign=1.0, scale=scale)
    return y


def record_fdfp(rec: Recorder, pfx: str, x: torch.Tensor, dim: int, hid: int, dims):
    """Record FDFP.forward in Fourier mode (block.py:532-554): Re fft2 -> conv1x1 + bias -> GELU -> conv1x1 + bias -> Re ifft2."""
    B, H, W = dims
    M, C_ = B * H * W, dim
    P, Gr = (lambda n: rec.P(pfx + n)), (lambda n: rec.Gr(pfx + n))
    W1, b1, W2, b2 = P("conv1.weight"), P("conv1.bias"), P("conv2.weight"), P("conv2.bias")
    n = pfx
    f = record_spatial_refft(rec, n + "fft2", x, dims, C_, 1.0)
    pre, t, z = rec.alloc(M, hid), rec.alloc(M, hid), rec.alloc(M, C_)
    rec.gemm(n + "conv1+gelu", f, W1, t, N=hid, K=C_, w_sn=C_, w_sk=1, bias=b1, act=L.ACT_GELU, pre_out=pre)
    rec.gemm(n + "conv2", t, W2, z, N=C_, K=hid, w_sn=hid, w_sk=1, bias=b2)
    y = record_spatial_refft(rec, n + "ifft2", z, dims, C_, 1.0 / (H * W))

    def backward(dy: torch.Tensor) -> torch.Tensor:
        dz = record_spatial_refft(rec, n + "ifft2.bwd", dy, dims, C_, 1.0 / (H * W))
        dt, dpre, df = rec.alloc(M, hid), rec.alloc(M, hid), rec.alloc(M, C_)
        with rec.bulk():
            rec.gram(n + "conv2.dW", dz, t, Gr("conv2.weight"), N=C_, K=hid, o_sn=hid, o_sk=1)
            _colsum_wide(rec, n + "conv2.db", dz, C_, Gr("conv2.bias"))
        rec.gemm(n + "conv2.dX", dz, W2, dt, N=hid, K=C_, w_sn=1, w_sk=hid)
        _act(rec, n + "gelu.bwd", pre, None, hid, bwd_dt=dt, dpre=dpre, dbias=Gr("conv1.bias"))
        with rec.bulk():
            rec.gram(n + "conv1.dW", dpre, f, Gr("conv1.weight"), N=hid, K=C_, o_sn=C_, o_sk=1)
        rec.gemm(n + "conv1.dX", dpre, W1, df, N=C_, K=hid, w_sn=1, w_sk=C_)
        return record_spatial_refft(rec, n + "fft2.bwd", df, dims, C_, 1.0)
    return y, backward


class TokenPlan(Recorder):
    """A launch plan over token rows [B*H*W, dim]: input buffer -> recorded forward -> output buffer, and the recorded backward.
    Subclasses provide ``_record(x) -> (y, backward)``."""

    def __init__(self, fp: FlatParams, dim: int, B: int, H: int, W: int, want_backward: bool = True):
        super().__init__(fp.device, fp, scratch_floats=8 * 1024 * 1024)
        self.dim, self.B, self.H, self.W = dim, B, H, W
        self.want_backward = want_backward
        M = B * H * W
        self.x_in = self.alloc(M, dim)
        self.cur = self.fwd
        self.y_out, bw = self._record(self.x_in)
        self.mark("fwd.end")
        if want_backward:
            self.cur = self.bwd
            self.dy_in = self.alloc(M, dim)
            self.dx_out = bw(self.dy_in)
            self.mark("bwd.end")
        self.cur = self.fwd

    def _record(self, x):
        raise NotImplementedError

    def forward(self, x_rows: torch.Tensor) -> torch.Tensor:
        self.x_in.copy_(x_rows)
        self.run(self.fwd)
        return self.y_out

    def backward(self, dy_rows: torch.Tensor) -> torch.Tensor:
        self.dy_in.copy_(dy_rows)
        self.run(self.bwd)
        return self.dx_out

    # algorithmic work of one forward + backward (for the bench tools): every launch's operands once
    def flops(self) -> int:
        return sum(m[3] for m in self.meta[id(self.fwd)]) + sum(m[3] for m in self.meta[id(self.bwd)])

    def alg_bytes(self) -> int:
        return sum(m[2] for m in self.meta[id(self.fwd)]) + sum(m[2] for m in self.meta[id(self.bwd)])


class FrfnPlan(TokenPlan):
    """Launch lists + buffers of one FRFN for a fixed (B, H, W)."""

    def __init__(self, fp: FlatParams, dim: int, hidden: int, B: int, H: int, W: int, want_backward: bool = True):
        self.hidden = hidden
        super().__init__(fp, dim, B, H, W, want_backward)

    def _record(self, x):
        return record_frfn(self, "", x, self.dim, self.hidden, (self.B, self.H, self.W))


class LeffPlan(TokenPlan):
    def __init__(self, fp: FlatParams, dim: int, hidden: int, B: int, H: int, W: int, want_backward: bool = True):
        self.hidden = hidden
        super().__init__(fp, dim, B, H, W, want_backward)

    def _record(self, x):
        return record_leff(self, "", x, self.dim, self.hidden, (self.B, self.H, self.W))


class FdfpPlan(TokenPlan):
    def __init__(self, fp: FlatParams, dim: int, hidden: int, B: int, H: int, W: int, want_backward: bool = True):
        self.hidden = hidden
        super().__init__(fp, dim, B, H, W, want_backward)

    def _record(self, x):
        return record_fdfp(self, "", x, self.dim, self.hidden, (self.B, self.H, self.W))


class _TokenFunction(torch.autograd.Function):
    """One autograd node for a whole token-plan module: forward and backward are launch lists."""

    @staticmethod
    def forward(ctx, x, module, *params):
        B, Ltok, C_ = x.shape
        plan = module._plan_for(x)
        y = plan.forward(x.reshape(B * Ltok, C_)).clone().view(B, Ltok, C_)
        module._serial += 1
        ctx.module, ctx.plan, ctx.serial, ctx.shape = module, plan, module._serial, (B, Ltok, C_)
        return y

    @staticmethod
    def backward(ctx, dy):
        m = ctx.module
        if ctx.serial != m._serial:
            raise RuntimeError(f"{type(m).__name__}(HIP): a later forward overwrote the saved activations; call backward before "
                               "the next forward")
        B, Ltok, C_ = ctx.shape
        dx = ctx.plan.backward(dy.contiguous().view(B * Ltok, C_)).clone().view(B, Ltok, C_)
        fp = m._flat
        return (dx, None) + tuple(None if n in m._dead else fp.gview(n).clone() for n in m._names)


class TokenModule(nn.Module):
    """Shared host plumbing of the token-plan modules: parameters as views of one flat buffer, one plan per (B, H, W),
    the reference's ``forward(x: (B, L, C))`` contract with L = H*H."""

    _dead: frozenset = frozenset()        # parameters the reference never uses (their .grad stays None, as there)

    def _init_plumbing(self):
        self._flat: "FlatParams | None" = None
        self._plans: Dict[Tuple[int, int, int], TokenPlan] = {}
        self._names = [n for n, _ in self.named_parameters()]
        self._serial = 0

    def _make_plan(self, fp, B, H, W, want_backward) -> TokenPlan:
        raise NotImplementedError

    def _ensure_flat(self, device) -> FlatParams:
        named = list(self.named_parameters())
        ok = self._flat is not None and self._flat.device == torch.device(device) and \
            all(p.data_ptr() == self._flat.view(n).data_ptr() for n, p in named)
        if not ok:
            fp = FlatParams({n: tuple(p.shape) for n, p in named}, device)
            with torch.no_grad():
                for n, p in named:
                    v = fp.view(n)
                    v.copy_(p.data.to(device=device, dtype=torch.float32))
                    p.data = v
            self._flat = fp
            self._plans.clear()
        return self._flat

    def _plan_for(self, x: torch.Tensor, want_backward: bool = True) -> TokenPlan:
        fp = self._ensure_flat(x.device)
        B, Ltok, _ = x.shape
        hh = int(math.sqrt(Ltok))
        key = (B, hh, hh)
        plan = self._plans.get(key)
        if plan is None or (want_backward and not plan.want_backward):
            while len(self._plans) >= 2:
                self._plans.pop(next(iter(self._plans)))
            plan = self._make_plan(fp, B, hh, hh, want_backward)
            self._plans[key] = plan
        return plan

    def forward(self, x: torch.Tensor, mask=None) -> torch.Tensor:
        if x.dim() != 3 or x.shape[2] != self.dim:
            raise ValueError(f"expected (B, L, {self.dim}) tokens, got {tuple(x.shape)}")
        hh = int(math.sqrt(x.shape[1]))
        if hh * hh != x.shape[1]:
            raise ValueError("needs a square token grid (L = H*H), like the reference's int(math.sqrt(hw))")
        if x.device.type != "cuda":
            raise RuntimeError(f"{type(self).__name__}(HIP) runs on an AMD GPU ('cuda' device in PyTorch-ROCm); there is no CPU fallback")
        x = x.contiguous().float()
        self._ensure_flat(x.device)
        if torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in self.parameters())):
            return _TokenFunction.apply(x, self, *self.parameters())
        B, Ltok, C_ = x.shape
        plan = self._plan_for(x, want_backward=False)
        self._serial += 1
        return plan.forward(x.reshape(B * Ltok, C_)).clone().view(B, Ltok, C_)


class FRFN(TokenModule):
    """Drop-in for ``src.model.block.FRFN`` (block.py:249-282)."""

    def __init__(self, dim=32, hidden_dim=128, act_layer=nn.GELU, drop=0., use_eca=False):
        super().__init__()
        if act_layer is not nn.GELU:
            raise NotImplementedError("the HIP FRFN implements the reference's configuration (act_layer=nn.GELU)")
        if dim % 16 or hidden_dim % 4:
            raise NotImplementedError("dim must be a multiple of 16 (C/4 channels in float4 groups), hidden_dim of 4")
        # registration order == reference (block.py:252-260): identical RNG consumption under a seed
        self.linear1 = nn.Sequential(nn.Linear(dim, hidden_dim * 2), act_layer())
        self.dwconv = nn.Sequential(nn.Conv2d(hidden_dim, hidden_dim, groups=hidden_dim, kernel_size=3, stride=1, padding=1), act_layer())
        self.linear2 = nn.Sequential(nn.Linear(hidden_dim, dim))
        self.dim = dim
        self.hidden_dim = hidden_dim
        self.dim_conv = self.dim // 4
        self.dim_untouched = self.dim - self.dim_conv
        self.partial_conv3 = nn.Conv2d(self.dim_conv, self.dim_conv, 3, padding=1, stride=1, bias=False)
        self._init_plumbing()

    def _make_plan(self, fp, B, H, W, want_backward):
        return FrfnPlan(fp, self.dim, self.hidden_dim, B, H, W, want_backward)


class LeFF(TokenModule):
    """Drop-in for ``src.model.block.LeFF`` (block.py:223-247), the default feed-forward of EncoderBlock / AST."""

    def __init__(self, dim=32, hidden_dim=128, act_layer=nn.GELU, drop=0., use_eca=False):
        super().__init__()
        if act_layer is not nn.GELU:
            raise NotImplementedError("the HIP LeFF implements the reference's configuration (act_layer=nn.GELU)")
        if dim % 4 or hidden_dim % 4:
            raise NotImplementedError("dim and hidden_dim must be multiples of 4 (float4 channel groups)")
        self.linear1 = nn.Sequential(nn.Linear(dim, hidden_dim), act_layer())
        self.dwconv = nn.Sequential(nn.Conv2d(hidden_dim, hidden_dim, groups=hidden_dim, kernel_size=3, stride=1, padding=1), act_layer())
        self.linear2 = nn.Sequential(nn.Linear(hidden_dim, dim))
        self.dim = dim
        self.hidden_dim = hidden_dim
        self.eca = nn.Identity()
        self._init_plumbing()

    def _make_plan(self, fp, B, H, W, want_backward):
        return LeffPlan(fp, self.dim, self.hidden_dim, B, H, W, want_backward)


class FDFP(TokenModule):
    """Drop-in for ``src.model.block.FDFP`` (block.py:518-556) in the mode the reference ships (``use_dwt="Fourier"``):
    ``forward(x: (B, H, W, C)) -> (B, H, W, C)``.  Parameters ``conv1.{weight,bias}`` [hidden, C, 1, 1], ``conv2.*`` [C, hidden, 1, 1]
    (no DWT_2D / IDWT_2D sub-modules: those are only constructed for a truthy non-"Fourier" flag and hold no parameters the
    Fourier mode uses)."""

    def __init__(self, in_channels, hidden_channels, act_layer=nn.GELU, use_dwt="Fourier"):
        super().__init__()
        if use_dwt != "Fourier" or act_layer is not nn.GELU:
            raise NotImplementedError("the HIP FDFP implements the reference's shipped mode (use_dwt='Fourier', GELU)")
        if in_channels % 4 or hidden_channels % 4:
            raise NotImplementedError("channel counts must be multiples of 4")
        self.use_dwt = use_dwt
        # The reference builds DWT_2D / IDWT_2D for any truthy flag (block.py:522-524); in Fourier mode they are never called,
        # but their Haar buffers are part of the state_dict: kept, built like wave_modules.py:122-134,159-171 from PyWavelets'
        # published Haar filter bank (float32 products of 1/sqrt(2)).
        r = 0.7071067811865476
        lo, hi = torch.Tensor([r, r]), torch.Tensor([r, -r])            # dec_lo[::-1], dec_hi[::-1]
        self.dwt = nn.Module()
        for nm, (a, b) in (("w_ll", (lo, lo)), ("w_hl", (hi, lo)), ("w_lh", (lo, hi)), ("w_hh", (hi, hi))):
            self.dwt.register_buffer(nm, a.unsqueeze(0) * b.unsqueeze(1))
        rlo, rhi = torch.Tensor([r, r]), torch.Tensor([r, -r])           # rec_lo, rec_hi
        self.idwt = nn.Module()
        self.idwt.register_buffer("filters", torch.stack([rlo.unsqueeze(0) * rlo.unsqueeze(1), rlo.unsqueeze(0) * rhi.unsqueeze(1),
                                                          rhi.unsqueeze(0) * rlo.unsqueeze(1), rhi.unsqueeze(0) * rhi.unsqueeze(1)], dim=0))
        self.conv1 = nn.Conv2d(in_channels, hidden_channels, kernel_size=1, stride=1)
        self.conv2 = nn.Conv2d(hidden_channels, in_channels, kernel_size=1, stride=1)
        self.act = act_layer()
        self.in_channels = in_channels
        self.hidden_channels = hidden_channels
        self.dim = in_channels
        self._init_plumbing()

    def _make_plan(self, fp, B, H, W, want_backward):
        return FdfpPlan(fp, self.in_channels, self.hidden_channels, B, H, W, want_backward)

    def _plan_for(self, x, want_backward=True):            # x arrives as (B, L, C) tokens from forward() below
        return super()._plan_for(x, want_backward)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if x.dim() != 4 or x.shape[3] != self.in_channels:
            raise ValueError(f"expected (B, H, W, {self.in_channels}), got {tuple(x.shape)}")
        B, H, W, C_ = x.shape
        if H != W:
            raise NotImplementedError("square token images (the reference's H = W = sqrt(L))")
        return super().forward(x.reshape(B, H * W, C_)).view(B, H, W, C_)
