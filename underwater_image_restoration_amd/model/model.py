"""``EncoderBlock`` of the NewBigFRFNModel family on MI355X behind the reference's surface
(src/model/model.py:16-93): same constructor arguments, same ``state_dict`` names in the same registration order
(``norm1.*``, ``mlp.*``, ``norm2.*``, ``freq_mlp.*``), ``forward(x: (B, L, C), mask=None) -> (B, L, C)``.

Built for the configuration MyBigFRFNModel uses for its first level (model.py:481-489): ``token_mlp = freq_mlp = "frfn"``
(or "leff", the class default), ``use_dwt = "Fourier"``.  In that mode the reference computes

    x1 = mlp(norm1(x));  f = Re fftn(x1, dims (L, C));  f = freq_mlp(f);  f = Re ifftn(f, dims (L, C));  out = x + f + x1

(``norm2`` is evaluated and discarded, :62,72: its parameters keep ``grad = None`` here too).  ``Re fftn`` over the (L, C) axes
of a real tensor is the real-symmetric operator  T(x) = C_L x C_C - S_L x S_C  (cos / sin DFT matrices), and ``Re ifftn`` is
T / (L*C): one operator, also its own backward pass.  T runs as two uir_gemm (C axis) and a two-stage length-L transform
(L = H*W = N1*N2: uir_dft_axis over n1, uir_cmul_table twiddles, uir_dft_axis over n2, uir_transpose_rows); the two FRFNs
are ``block.record_frfn``; LayerNorm is uir_layernorm_fwd / uir_layernorm_bwd.  No CPU / eager fallback.

``token_mlp`` / ``freq_mlp`` may each be "frfn" or "leff" (the reference's default); ``drop_path > 0`` applies timm's
stochastic depth (per-sample factors, training mode only).  Not built (raise NotImplementedError): the Wavelet mode (the
deeper encoder levels), DecoderBlock / MDASSA / FDFP - SURVEY.md 8(f)-2, next.
"""
from __future__ import annotations

import os

import numpy as np
import torch
import torch.nn as nn

from .. import _lib as L
from ..engine import FlatParams, _ptr
from .block import (FRFN, LeFF, MDASSA, MDASSA_DEAD, _add_rows, Downsample, InputProjection, OutputProjection, TokenModule, TokenPlan, Upsample,
                    record_decoder_block, record_frfn, record_leff)
from .. import ops

lib = L.lib
_TABLES: dict = {}


def _axis_tables(C: int, device):
    """cos / sin(2 pi c' c / C), [C][C], fp64-built on exact residues."""
    key = ("C", C, str(device))
    if key not in _TABLES:
        r = (np.arange(C, dtype=np.int64)[:, None] * np.arange(C, dtype=np.int64)[None, :]) % C
        ang = 2.0 * np.pi * r.astype(np.float64) / C
        c, s = np.cos(ang), np.sin(ang)
        q = (4 * r) % C == 0
        c[q], s[q] = np.round(c[q]), np.round(s[q])
        _TABLES[key] = (torch.from_numpy(c.astype(np.float32)).to(device).contiguous(),
                        torch.from_numpy(s.astype(np.float32)).to(device).contiguous())
    return _TABLES[key]


def _twiddle_tables(N1: int, Mrest: int, device):
    """exp(+2 pi i k1 m / (N1*Mrest)) at p = k1*Mrest + m: the twiddle between a stage of length N1 and the rest."""
    key = ("tw", N1, Mrest, str(device))
    if key not in _TABLES:
        Ltok = N1 * Mrest
        r = (np.arange(N1, dtype=np.int64)[:, None] * np.arange(Mrest, dtype=np.int64)[None, :]) % Ltok
        ang = 2.0 * np.pi * r.astype(np.float64) / Ltok
        c, s = np.cos(ang), np.sin(ang)
        q = (4 * r) % Ltok == 0
        c[q], s[q] = np.round(c[q]), np.round(s[q])
        _TABLES[key] = (torch.from_numpy(c.astype(np.float32).reshape(-1)).to(device).contiguous(),
                        torch.from_numpy(s.astype(np.float32).reshape(-1)).to(device).contiguous())
    return _TABLES[key]


_FAST = (128, 64, 32)          # lengths the two-stage MFMA DFT kernel (uir_dft_axis2) handles; anything else is a dense contraction


def token_factors(H: int, W: int):
    """Stage lengths (N1, N2[, N3]) with product H*W.  Prefer factors the fast DFT kernel handles (65 536 = 64*32*32,
    4096 = 64*64); otherwise the two dense stages (H, W)."""
    Ltok = H * W
    for a in _FAST:
        if Ltok % a == 0 and Ltok // a in _FAST:
            return (a, Ltok // a)
    for a in _FAST:
        for b in _FAST:
            if Ltok % (a * b) == 0 and Ltok // (a * b) in _FAST and a >= b >= Ltok // (a * b):
                return (a, b, Ltok // (a * b))
    return (H, W)


def record_token_fft(rec, name: str, x: torch.Tensor, dims, C_: int, scale: float) -> torch.Tensor:
    """y = scale * Re(sum_{l,c} x[l,c] exp(i 2 pi (k l / L + c' c / C))) = scale * (C_L x C_C - S_L x S_C) on token rows."""
    B, H, W = dims
    M, Ltok = B * H * W, H * W
    f = token_factors(H, W)                # token l = (n1 * N2 + n2) * N3 + n3
    Cc, Sc = _axis_tables(C_, rec.dev)
    rec._hold(Cc, Sc)
    zr, zi = rec.alloc(M, C_), rec.alloc(M, C_)
    rec.gemm(name + ".cosC", x, Cc, zr, N=C_, K=C_, w_sn=C_, w_sk=1)
    rec.gemm(name + ".sinC", x, Sc, zi, N=C_, K=C_, w_sn=C_, w_sk=1)
    outer, rest = B, Ltok
    for si, n in enumerate(f):
        rest //= n
        last = si == len(f) - 1
        ar = rec.alloc(M, C_)
        ai = None if last else rec.alloc(M, C_)
        rec.dft(f"{name}.n{si + 1}", zr, zi, ar, ai, O=outer, N=n, I=rest, C_=C_, sign=1.0, scale=1.0)
        if not last:       # twiddle between this stage and the remaining length-(n * rest) ... (rest) transform
            twr, twi = _twiddle_tables(n, rest, rec.dev)
            rec._hold(twr, twi)
            rec._op(f"{name}.twiddle{si + 1}", lib.uir_cmul_table, _ptr(ar), _ptr(ai), _ptr(twr), _ptr(twi), M, n * rest, C_,
                    nbytes=16 * M * C_)
        zr, zi = ar, ai
        outer *= n
    y = rec.alloc(M, C_)
    n1, n2, n3 = (f[0], 1, f[1]) if len(f) == 2 else f            # stage-ordered frequency index -> natural order
    rec._op(name + ".natural", lib.uir_transpose_rows, _ptr(zr), _ptr(y), B, n1, n2, n3, C_, scale, nbytes=8 * M * C_)
    return y


_MLP_RECORDERS = {"frfn": record_frfn, "leff": record_leff}


class EncoderPlan(TokenPlan):
    def __init__(self, fp: FlatParams, dim: int, hidden: int, B: int, H: int, W: int, want_backward: bool = True,
                 token_mlp: str = "frfn", freq_mlp: str = "frfn"):
        self.hidden, self.token_mlp, self.freq_mlp = hidden, token_mlp, freq_mlp
        super().__init__(fp, dim, B, H, W, want_backward)

    def _record(self, x):
        C_, dims = self.dim, (self.B, self.H, self.W)
        M, Ltok = self.B * self.H * self.W, self.H * self.W
        g1, b1 = self.P("norm1.weight"), self.P("norm1.bias")
        xn, stats = self.alloc(M, C_), self.alloc(M, 2)
        self._hold(g1, b1)
        self._op("norm1", lib.uir_layernorm_fwd, _ptr(x), C_, _ptr(g1), _ptr(b1), _ptr(xn), C_, _ptr(stats), M, C_, 1e-5,
                 nbytes=8 * M * C_)
        x1, bw1 = _MLP_RECORDERS[self.token_mlp](self, "mlp.", xn, C_, self.hidden, dims, owns_input=True)     # xn, f1: private intermediates
        f1 = record_token_fft(self, "fftn", x1, dims, C_, 1.0)
        f2, bw2 = _MLP_RECORDERS[self.freq_mlp](self, "freq_mlp.", f1, C_, self.hidden, dims, owns_input=True)
        f3 = record_token_fft(self, "ifftn", f2, dims, C_, 1.0 / (Ltok * C_))
        # x = shortcut + drop_path2(freq_x) + drop_path(x)  (model.py:91): per-sample factors (1 in eval mode / for drop_path 0)
        self.dp_scale = self.alloc(2, self.B)          # row 0: drop_path2 (frequency branch), row 1: drop_path (token branch)
        self.dp_scale.fill_(1.0)
        s2, s1 = self.dp_scale[0], self.dp_scale[1]
        y0 = _add_rows(self, "out=x+dp2*freq", x, f3, C_, scale=s2, rows_per_sample=Ltok)
        y = _add_rows(self, "out+=dp*x1", y0, x1, C_, scale=s1, rows_per_sample=Ltok)

        def backward(dy):
            dyf = self.alloc(M, C_)
            self._op("dfreq=dp2*dy", lib.uir_scale_rows, _ptr(dy), C_, _ptr(dyf), C_, _ptr(s2), Ltok, M, C_, 0, nbytes=8 * M * C_)
            g2 = record_token_fft(self, "ifftn.bwd", dyf, dims, C_, 1.0 / (Ltok * C_))    # T is symmetric: its own adjoint
            df1 = bw2(g2)
            dx1 = record_token_fft(self, "fftn.bwd", df1, dims, C_, 1.0)
            self._op("dx1+=dp*dy", lib.uir_scale_rows, _ptr(dy), C_, _ptr(dx1), C_, _ptr(s1), Ltok, M, C_, 1, nbytes=12 * M * C_)
            dxn = bw1(dx1)
            dx = self.alloc(M, C_)
            self.ln_bwd("norm1.bwd", dxn, x, stats, g1, dy, dx, self.Gr("norm1.weight"), self.Gr("norm1.bias"), C_=C_)
            return dx
        return y, backward


class EncoderBlock(TokenModule):
    """Drop-in for ``src.model.model.EncoderBlock`` (model.py:16-93) in the mode MyBigFRFNModel's first level uses."""

    _dead = frozenset(("norm2.weight", "norm2.bias"))

    def __init__(self, dim, input_resolution, num_heads, mlp_ratio=4, token_mlp="leff", drop_path=0.0, norm_layer=nn.LayerNorm,
                 act_layer=nn.GELU, drop=0.0, freq_mlp="leff", use_dwt="Fourier"):
        super().__init__()
        if token_mlp not in ("leff", "frfn"):
            raise ValueError(f"Unknown token_mlp type: {token_mlp}")                     # the reference's errors (model.py:38,48)
        if freq_mlp not in ("leff", "frfn"):
            raise ValueError(f"Unknown freq_mlp type: {freq_mlp}")
        if use_dwt != "Fourier" or norm_layer is not nn.LayerNorm or act_layer is not nn.GELU:
            raise NotImplementedError("the HIP EncoderBlock is built for use_dwt='Fourier', LayerNorm, GELU (MyBigFRFNModel, model.py:481-524)")
        if not 0.0 <= drop_path < 1.0:
            raise ValueError("drop_path must be in [0, 1)")
        if dim not in (32, 64, 128, 256, 512):
            raise NotImplementedError("dim in {32, 64, 128, 256, 512} (the LayerNorm kernels' channel counts)")
        self.dim = dim
        self.input_resolution = input_resolution
        self.num_heads = num_heads
        self.mlp_ratio = mlp_ratio
        self.token_mlp = token_mlp
        self.drop_prob = float(drop_path)
        self.drop_path = nn.Identity()                   # the stochastic-depth factors are applied inside the launch plan
        self.norm1 = norm_layer(dim)
        self.use_dwt = use_dwt
        mlp_hidden_dim = int(dim * mlp_ratio)
        self.hidden_dim = mlp_hidden_dim
        mlps = {"frfn": FRFN, "leff": LeFF}                                             # parameter containers in the reference's
        self.mlp = mlps[token_mlp](dim, mlp_hidden_dim, act_layer=act_layer, drop=drop)  # registration order (model.py:28-52)
        self.norm2 = norm_layer(dim)
        self.freq_mlp = mlps[freq_mlp](dim, mlp_hidden_dim, act_layer=act_layer, drop=drop)
        self.freq_mlp_kind = freq_mlp
        self.drop_path2 = nn.Identity()
        self._init_plumbing()

    def _make_plan(self, fp, B, H, W, want_backward):
        return EncoderPlan(fp, self.dim, self.hidden_dim, B, H, W, want_backward, self.token_mlp, self.freq_mlp_kind)

    def _before_forward(self, plan) -> None:
        """timm's drop_path (the published function the reference imports, model.py:7,27): in training mode with p > 0 each of
        the two branches gets an independent per-sample factor Bernoulli(1 - p) / (1 - p); drawn in the reference's order
        (``shortcut + drop_path2(freq_x) + drop_path(x)``: drop_path2 first) from torch's device generator."""
        if self.training and self.drop_prob > 0.0:
            keep = 1.0 - self.drop_prob
            for r in (0, 1):
                plan.dp_scale[r].copy_(torch.empty(plan.B, device=plan.dp_scale.device).bernoulli_(keep).div_(keep))
        else:
            plan.dp_scale.fill_(1.0)


class DecoderPlan(TokenPlan):
    def __init__(self, fp: FlatParams, dim: int, mdim: int, heads: int, token_mlp: str, B: int, H: int, W: int, want_backward: bool = True):
        self.out_width, self.heads, self.token_mlp = dim, heads, token_mlp
        super().__init__(fp, mdim, B, H, W, want_backward, out_dim=dim)

    def _record(self, x):
        return record_decoder_block(self, "", x, self.out_width, self.dim, self.heads, self.token_mlp, (self.B, self.H, self.W))


class DecoderBlock(TokenModule):
    """Drop-in for ``src.model.model.DecoderBlock`` (model.py:96-160) as MyBigFRFNModel builds it (win_size 8, shift_size 0,
    drop_path 0, Fourier): ``forward(x: (B, L, dim), enc_out: (B, L, dim) | None) -> (B, L, dim)``.  With ``enc_out=True`` the
    block works on the concatenation (2*dim channels) and projects back to dim (``mlp_proj``)."""

    def __init__(self, dim, input_resolution, num_heads, win_size=8, shift_size=0, mlp_ratio=4, token_mlp="leff", drop_path=0.0,
                 norm_layer=nn.LayerNorm, act_layer=nn.GELU, drop=0.0, token_projection="linear", enc_out=True, freq_attn_win_ratio=2,
                 use_dwt="Fourier"):
        super().__init__()
        if token_mlp not in ("leff", "frfn"):
            raise ValueError(f"Unknown token_mlp type: {token_mlp}")                     # model.py:136
        if drop_path > 0.0 or shift_size != 0 or mlp_ratio != 4:
            raise NotImplementedError("the HIP DecoderBlock is built for drop_path 0, shift_size 0, mlp_ratio 4 (MyBigFRFNModel)")
        self.dim, self.input_resolution, self.num_heads, self.mlp_ratio = dim, input_resolution, num_heads, mlp_ratio
        self.token_mlp, self.enc_out, self.win_size, self.shift_size = token_mlp, enc_out, win_size, shift_size
        if min(self.input_resolution) <= self.win_size:
            raise NotImplementedError("token images larger than one 8 x 8 window")
        self.drop_path = nn.Identity()
        mdim = dim * 2 if enc_out else dim
        self.mdim, self._out_width = mdim, dim
        self.norm1 = norm_layer(mdim)                                                    # registration order: model.py:119-139
        self.norm2 = norm_layer(mdim)
        self.mdassa = MDASSA(mdim, num_heads=num_heads, win_size=win_size, shift_size=shift_size, token_projection=token_projection,
                             qkv_bias=True, qk_scale=None, attn_drop=0., proj_drop=0., norm_layer=norm_layer, enc_out=enc_out,
                             freq_attn_win_ratio=freq_attn_win_ratio, use_dwt=use_dwt)
        self.mlp = (FRFN if token_mlp == "frfn" else LeFF)(mdim, int(mdim * mlp_ratio), act_layer=act_layer, drop=drop)
        self.mlp_proj = nn.Linear(mdim, dim)
        self._dead = frozenset("mdassa." + n + s for n in MDASSA_DEAD for s in ("weight", "bias"))
        self._init_plumbing()

    def _make_plan(self, fp, B, H, W, want_backward):
        return DecoderPlan(fp, self._out_width, self.mdim, self.num_heads, self.token_mlp, B, H, W, want_backward)

    def forward(self, x: torch.Tensor, enc_out=None) -> torch.Tensor:
        if (enc_out is not None) != bool(self.enc_out):
            raise ValueError("enc_out must be given exactly when the block was built with enc_out=True")
        if enc_out is not None:
            x = torch.cat([x, enc_out], dim=2)                                           # model.py:146
        d, self.dim = self.dim, self.mdim                                                # TokenModule.forward checks the input width
        try:
            return super().forward(x)
        finally:
            self.dim = d


class _ResidualImage(torch.autograd.Function):
    """out + x for two NCHW images (MyBigFRFNModel.forward: ``out = out + x``, model.py:640) through uir_copy_cols; x is the input
    image and gets no gradient."""

    @staticmethod
    def forward(ctx, out, x):
        y = out.clone()
        n = y.numel()
        if n % 4:
            raise ValueError("image element count must be a multiple of 4")
        ops.flat_accumulate(x.contiguous().view(-1), y.view(-1), n, accumulate=True)
        return y

    @staticmethod
    def backward(ctx, g):
        return g, None


class MyBigFRFNModel(nn.Module):
    """``--arch NewBigFRFNModel`` (src/model/model.py:465-643): a 5-level U-Net of EncoderBlocks (FRFN + (L, C)-Fourier branch) and
    DecoderBlocks (MDASSA window attention + FRFN) over token images.  Same constructor, child names and registration order
    (hence ``state_dict`` keys) as the reference.  Every child is a launch-plan module of this package; tensors between them are
    device tensors handed over through autograd (one node per child).  Differences, both documented in DESIGN.md section 14:
    ``output_proj`` receives the token image (the reference hands its NCHW-only OutputProjection the tokens and crashes there,
    SURVEY.md fact 2; the reshape of model.py:435-437 is what is meant), and the per-forward ``print`` of the bottleneck shape
    (:628) is not emitted."""

    # device memory the 27 launch plans keep per input pixel (measured, tests/test_frfn_net.py holds it): ModuleTrainStep sizes micro-batches with it
    # (tools/measure_module_plans.py, round 3 with the activations of FRFN fused away: 4.0 / 6.1 / 19.5 GiB at 1 x 128^2 / 2 x 128^2 / 2 x 256^2 = 147 KB per pixel + 1.7 GB of per-plan scratch)
    PLAN_BYTES_PER_PIXEL_TRAIN, PLAN_BYTES_FIXED_TRAIN = 147_000, 1_900_000_000
    PLAN_BYTES_PER_PIXEL_EVAL, PLAN_BYTES_FIXED_EVAL = 69_000, 1_750_000_000


    def __init__(self, img_size=512, dd_in=3, embed_dim=32, dropout_rate=0., drop_path_rate=0.1, use_dwt="Fourier"):
        super().__init__()
        if dd_in != 3 or dropout_rate != 0. or use_dwt != "Fourier":
            raise NotImplementedError("the HIP MyBigFRFNModel is built for dd_in=3, dropout_rate=0, use_dwt='Fourier' (the defaults)")
        self.img_size, self.embed_dim, self.num_enc_layers = img_size, embed_dim, 4
        E_, r = embed_dim, img_size
        self.input_proj = InputProjection(in_channels=dd_in, out_channels=E_)
        self.pos_drop = nn.Dropout(p=dropout_rate)
        dpr = [x.item() for x in torch.linspace(0, drop_path_rate, self.num_enc_layers)]

        def enc(dim, res, dp):
            return EncoderBlock(dim=dim, input_resolution=(res, res), num_heads=4, mlp_ratio=4, token_mlp="frfn", drop_path=dp,
                                norm_layer=nn.LayerNorm, act_layer=nn.GELU, drop=dropout_rate, freq_mlp="frfn", use_dwt=use_dwt)

        def dec(dim, res, enc_out, ratio):
            return DecoderBlock(dim=dim, input_resolution=(res, res), num_heads=4, win_size=8, shift_size=0, mlp_ratio=4,
                                token_mlp="frfn", drop_path=0.0, norm_layer=nn.LayerNorm, act_layer=nn.GELU, drop=dropout_rate,
                                token_projection="linear", enc_out=enc_out, freq_attn_win_ratio=ratio, use_dwt=use_dwt)
        # registration order == reference (model.py:474-583); the *_1 encoders all take enc_dpr[0] there
        self.encoder_0, self.encoder_0_1 = enc(E_, r, dpr[0]), enc(E_, r, dpr[0])
        self.downsample_0 = Downsample(E_, E_ * 2)
        self.encoder_1, self.encoder_1_1 = enc(E_ * 2, r // 2, dpr[1]), enc(E_ * 2, r // 2, dpr[0])
        self.downsample_1 = Downsample(E_ * 2, E_ * 4)
        self.encoder_2, self.encoder_2_1 = enc(E_ * 4, r // 4, dpr[2]), enc(E_ * 4, r // 4, dpr[0])
        self.downsample_2 = Downsample(E_ * 4, E_ * 8)
        self.encoder_3, self.encoder_3_1 = enc(E_ * 8, r // 8, dpr[3]), enc(E_ * 8, r // 8, dpr[0])
        self.downsample_3 = Downsample(E_ * 8, E_ * 16)
        self.bottleneck = dec(E_ * 16, r // 16, False, 2)
        self.upsample_3 = Upsample(E_ * 16, E_ * 8)
        self.decoder_3, self.decoder_3_1 = dec(E_ * 8, r // 8, True, 2), dec(E_ * 8, r // 8, False, 2)
        self.upsample_2 = Upsample(E_ * 8, E_ * 4)
        self.decoder_2, self.decoder_2_1 = dec(E_ * 4, r // 4, True, 4), dec(E_ * 4, r // 4, False, 2)
        self.upsample_1 = Upsample(E_ * 4, E_ * 2)
        self.decoder_1, self.decoder_1_1 = dec(E_ * 2, r // 2, True, 8), dec(E_ * 2, r // 2, False, 2)
        self.upsample_0 = Upsample(E_ * 2, E_)
        self.decoder_0, self.decoder_0_1 = dec(E_, r, True, 16), dec(E_, r, False, 2)
        self.output_proj = OutputProjection(in_channels=E_, out_channel=dd_in, kernel_size=3, stride=1, norm_layer=None, act_layer=None)

    def forward(self, x, mask=None):
        if mask is not None:
            raise NotImplementedError("mask=None (the trainer never passes one, src/ModelTrainer.py:79)")
        if x.dim() != 4 or x.shape[1] != 3 or x.shape[2] != x.shape[3] or x.shape[2] % 128:
            raise ValueError(f"expected (B, 3, S, S) with S a multiple of 128 (four 2x down-samplings to at least one 8 x 8 window), got {tuple(x.shape)}")
        y = self.input_proj(x)
        conv0 = self.encoder_0_1(self.encoder_0(y))
        conv1 = self.encoder_1_1(self.encoder_1(self.downsample_0(conv0)))
        conv2 = self.encoder_2_1(self.encoder_2(self.downsample_1(conv1)))
        conv3 = self.encoder_3_1(self.encoder_3(self.downsample_2(conv2)))
        bottleneck = self.bottleneck(self.downsample_3(conv3))
        dec3 = self.decoder_3_1(self.decoder_3(self.upsample_3(bottleneck), enc_out=conv3))
        dec2 = self.decoder_2_1(self.decoder_2(self.upsample_2(dec3), enc_out=conv2))
        dec1 = self.decoder_1_1(self.decoder_1(self.upsample_1(dec2), enc_out=conv1))
        dec0 = self.decoder_0_1(self.decoder_0(self.upsample_0(dec1), enc_out=conv0))
        out = self.output_proj(dec0)
        return _ResidualImage.apply(out, x.contiguous().float())

    def plan_modules(self):
        """The launch-plan children, in registration order (each owns one flat parameter / gradient buffer)."""
        return [m for m in self.children() if isinstance(m, TokenModule)]
