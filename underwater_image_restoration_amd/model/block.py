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


class FrfnPlan(Recorder):
    """Launch lists + buffers of one FRFN for a fixed (B, H, W)."""

    def __init__(self, fp: FlatParams, dim: int, hidden: int, B: int, H: int, W: int, want_backward: bool = True):
        super().__init__(fp.device, fp, scratch_floats=8 * 1024 * 1024)
        self.dim, self.hidden, self.B, self.H, self.W = dim, hidden, B, H, W
        self.want_backward = want_backward
        self._build()

    def _act(self, name, pre, t, cols, bwd_dt=None, dpre=None):
        self._hold(pre, t, bwd_dt, dpre)
        M = pre.shape[0]
        if bwd_dt is None:
            self._op(name, lib.uir_frfn_act_fwd, _ptr(pre), _ld(pre), _ptr(t), _ld(t), M, cols, nbytes=8 * M * cols)
        else:
            self._op(name, lib.uir_frfn_act_bwd, _ptr(pre), _ld(pre), _ptr(bwd_dt), _ld(bwd_dt), _ptr(dpre), _ld(dpre), M, cols,
                     nbytes=12 * M * cols)

    def _gate(self, name, U, bias, V, G=None, dG=None, dU=None, dV=None):
        self._hold(U, bias, V, G, dG, dU, dV)
        a = L.FrfnGateArgs()
        a.U, a.bias, a.V = _ptr(U), _ptr(bias), _ptr(V)
        a.M, a.hid = U.shape[0], self.hidden
        a.ldu, a.ldv = _ld(U), _ld(V)
        if dG is None:
            a.G, a.ldg = _ptr(G), _ld(G)
            self.keep.append(a)
            self._op(name, lib.uir_frfn_gate_fwd, C.byref(a), nbytes=12 * a.M * a.hid)
        else:
            a.dG, a.ldg = _ptr(dG), _ld(dG)
            a.dU, a.dV, a.lddu, a.lddv = _ptr(dU), _ptr(dV), _ld(dU), _ld(dV)
            self.keep.append(a)
            self._op(name, lib.uir_frfn_gate_bwd, C.byref(a), nbytes=20 * a.M * a.hid)

    def _colsum_wide(self, name, src, cols, out):
        """Bias gradient over `cols` columns: uir_colsum_partials handles <= 256 columns per launch."""
        for c0 in range(0, cols, 256):
            n = min(256, cols - c0)
            self.colsum(f"{name}[{c0}]", src, c0, n, out[c0:c0 + n])

    def _build(self):
        C_, hid, B, H, W = self.dim, self.hidden, self.B, self.H, self.W
        M, dims, c4 = B * H * W, (B, H, W), self.dim // 4
        P, Gr = self.P, self.Gr
        W1, b1 = P("linear1.0.weight"), P("linear1.0.bias")
        Wdw, bdw = P("dwconv.0.weight"), P("dwconv.0.bias")
        W2, b2 = P("linear2.0.weight"), P("linear2.0.bias")
        Wpc = P("partial_conv3.weight")
        self.x_in, self.y_out = self.alloc(M, C_), self.alloc(M, C_)
        x = self.x_in
        xc, pre, t = self.alloc(M, C_), self.alloc(M, 2 * hid), self.alloc(M, 2 * hid)
        u, g = self.alloc(M, hid), self.alloc(M, hid)
        self.cur = self.fwd
        # partial_conv3 on channels [0, C/4), the rest passes through (block.py:267-270)
        self.copy_cols("pc3.pass", x, c4, xc, c4, C_ - c4)
        self.gemm("pc3", x[:, :c4], Wpc, xc[:, :c4], N=c4, K=c4, w_sn=c4 * 9, w_sk=9, w_st=1, taps=9, hw=(H, W))
        self.gemm("linear1", xc, W1, pre, N=2 * hid, K=C_, w_sn=C_, w_sk=1, bias=b1)
        self._act("linear1.gelu", pre, t, 2 * hid)
        self.dwconv("dwconv", t[:, :hid], Wdw, u, C_=hid, dims=dims)
        self._gate("dwconv.gelu*x2", u, bdw, t[:, hid:], G=g)
        self.gemm("linear2", g, W2, self.y_out, N=C_, K=hid, w_sn=hid, w_sk=1, bias=b2)
        self.mark("fwd.end")
        if not self.want_backward:
            return
        self.cur = self.bwd
        self.dy_in, self.dx_out = self.alloc(M, C_), self.alloc(M, C_)
        dy = self.dy_in
        dg, du, dt = self.alloc(M, hid), self.alloc(M, hid), self.alloc(M, 2 * hid)
        dpre, dxc = self.alloc(M, 2 * hid), self.alloc(M, C_)
        with self.bulk():
            self.gram("linear2.dW", dy, g, Gr("linear2.0.weight"), N=C_, K=hid, o_sn=hid, o_sk=1)
            self._colsum_wide("linear2.db", dy, C_, Gr("linear2.0.bias"))
        self.gemm("linear2.dX", dy, W2, dg, N=hid, K=C_, w_sn=1, w_sk=hid)
        self._gate("gate.bwd", u, bdw, t[:, hid:], dG=dg, dU=du, dV=dt[:, hid:])
        with self.bulk():
            self._colsum_wide("dwconv.db", du, hid, Gr("dwconv.0.bias"))
        self.dwconv_bwd("dwconv.bwd", du, t[:, :hid], Wdw, dt[:, :hid], Gr("dwconv.0.weight"), C_=hid, creal=hid, dims=dims)
        self._act("linear1.gelu.bwd", pre, None, 2 * hid, bwd_dt=dt, dpre=dpre)
        with self.bulk():
            self.gram("linear1.dW", dpre, xc, Gr("linear1.0.weight"), N=2 * hid, K=C_, o_sn=C_, o_sk=1)
            self._colsum_wide("linear1.db", dpre, 2 * hid, Gr("linear1.0.bias"))
        self.gemm("linear1.dX", dpre, W1, dxc, N=C_, K=2 * hid, w_sn=1, w_sk=C_)
        self.copy_cols("pc3.pass.bwd", dxc, c4, self.dx_out, c4, C_ - c4)
        with self.bulk():
            self.gram("pc3.dW", dxc[:, :c4], x[:, :c4], Gr("partial_conv3.weight"), N=c4, K=c4, o_sn=c4 * 9, o_sk=9, o_st=1, taps=9,
                      hw=(H, W))
        self.gemm("pc3.dX", dxc[:, :c4], Wpc, self.dx_out[:, :c4], N=c4, K=c4, w_sn=9, w_sk=c4 * 9, w_st=1, taps=9, flip=1, hw=(H, W))
        self.mark("bwd.end")
        self.cur = self.fwd

    def forward(self, x_rows: torch.Tensor) -> torch.Tensor:
        self.x_in.copy_(x_rows)
        self.run(self.fwd)
        return self.y_out

    def backward(self, dy_rows: torch.Tensor) -> torch.Tensor:
        self.dy_in.copy_(dy_rows)
        self.run(self.bwd)
        return self.dx_out

    # algorithmic work of one forward + backward (for the bench): every leaf op's operands once
    def flops(self) -> int:
        return sum(m[3] for m in self.meta[id(self.fwd)]) + sum(m[3] for m in self.meta[id(self.bwd)])

    def alg_bytes(self) -> int:
        return sum(m[2] for m in self.meta[id(self.fwd)]) + sum(m[2] for m in self.meta[id(self.bwd)])


class _FrfnFunction(torch.autograd.Function):
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
            raise RuntimeError("FRFN(HIP): a later forward overwrote the saved activations; call backward before the next forward")
        B, Ltok, C_ = ctx.shape
        dx = ctx.plan.backward(dy.contiguous().view(B * Ltok, C_)).clone().view(B, Ltok, C_)
        fp = m._flat
        return (dx, None) + tuple(fp.gview(n).clone() for n in m._names)


class FRFN(nn.Module):
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
        self._flat: "FlatParams | None" = None
        self._plans: Dict[Tuple[int, int, int], FrfnPlan] = {}
        self._names = [n for n, _ in self.named_parameters()]
        self._serial = 0

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

    def _plan_for(self, x: torch.Tensor, want_backward: bool = True) -> FrfnPlan:
        fp = self._ensure_flat(x.device)
        B, Ltok, _ = x.shape
        hh = int(math.sqrt(Ltok))
        key = (B, hh, hh)
        plan = self._plans.get(key)
        if plan is None or (want_backward and not plan.want_backward):
            while len(self._plans) >= 2:
                self._plans.pop(next(iter(self._plans)))
            plan = FrfnPlan(fp, self.dim, self.hidden_dim, B, hh, hh, want_backward)
            self._plans[key] = plan
        return plan

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if x.dim() != 3 or x.shape[2] != self.dim:
            raise ValueError(f"expected (B, L, {self.dim}) tokens, got {tuple(x.shape)}")
        hh = int(math.sqrt(x.shape[1]))
        if hh * hh != x.shape[1]:
            raise ValueError("FRFN needs a square token grid (L = H*H), like the reference's int(math.sqrt(hw))")
        if x.device.type != "cuda":
            raise RuntimeError("FRFN(HIP) runs on an AMD GPU ('cuda' device in PyTorch-ROCm); there is no CPU fallback")
        x = x.contiguous().float()
        self._ensure_flat(x.device)
        if torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in self.parameters())):
            return _FrfnFunction.apply(x, self, *self.parameters())
        B, Ltok, C_ = x.shape
        plan = self._plan_for(x, want_backward=False)
        self._serial += 1
        return plan.forward(x.reshape(B * Ltok, C_)).clone().view(B, Ltok, C_)
