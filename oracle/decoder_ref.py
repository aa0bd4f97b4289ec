"""CPU oracle of the reference's window attention, MDASSA and DecoderBlock - TEST INFRASTRUCTURE ONLY.

Functional restatement, on the reference's parameter names, of
  WindowAttention_Sparse.forward   src/model/block.py:336-373   (q scale, + relative position bias, softmax * w1 + relu^2 * w2, proj)
  LinearProjection.forward         :166-200                      (to_q; to_kv_from_q for self attention, to_kv for attn_kv)
  window_partition / _reverse      :16-41                        (dilation 1)
  MDASSA.forward                   :409-515                      (shift_size 0, mask None; norm_q / norm_kv are commented out there)
  DecoderBlock.forward             src/model/model.py:142-160
for the configuration MyBigFRFNModel builds (win_size 8, heads 4, Fourier FDFP, no dropout / DropPath).
Pinned by tests/golden/decoder_ref.npz (outputs and gradients of the reference's own classes, tests/golden/make_decoder_vectors.py).
"""
from __future__ import annotations

from typing import Dict

import numpy as np
import torch
import torch.nn.functional as F

from .cpu_ref import _hash_uniform, _salt
from . import frfn_ref as R

Params = Dict[str, torch.Tensor]
WS = 8


def closed_form_params(shapes: "Dict[str, tuple]", salt: str, dtype=torch.float32) -> Params:
    """Deterministic values for every floating-point parameter of a module, by name and shape (no RNG)."""
    P = {}
    for n, shp in shapes.items():
        numel = int(np.prod(shp))
        h = _hash_uniform(numel, _salt(salt + "." + n) + numel % 97).reshape(shp)
        if n.endswith("relative_position_bias_table"):
            v = 0.5 * h
        elif n.endswith(".w") or n == "w":
            v = 0.3 * h
        elif "norm" in n.split(".")[-2] if "." in n else False:
            v = 1.0 + 0.2 * h if n.endswith("weight") else 0.1 * h
        elif len(shp) == 1:
            v = 0.1 * h
        else:
            v = h * (int(np.prod(shp[1:])) ** -0.5)
        P[n] = torch.from_numpy(np.ascontiguousarray(v)).to(dtype)
    return P


def relative_position_index(ws: int = WS) -> torch.Tensor:
    c = torch.stack(torch.meshgrid([torch.arange(ws), torch.arange(ws)], indexing="ij")).flatten(1)
    rel = (c[:, :, None] - c[:, None, :]).permute(1, 2, 0).contiguous()
    rel[:, :, 0] += ws - 1
    rel[:, :, 1] += ws - 1
    rel[:, :, 0] *= 2 * ws - 1
    return rel.sum(-1)


def _partition(t: torch.Tensor) -> torch.Tensor:
    B, H, W, c = t.shape
    return t.reshape(B, H // WS, WS, W // WS, WS, c).permute(0, 1, 3, 2, 4, 5).reshape(-1, WS * WS, c)


def _reverse(t: torch.Tensor, B: int, H: int, W: int) -> torch.Tensor:
    c = t.shape[-1]
    return t.reshape(B, H // WS, W // WS, WS, WS, c).permute(0, 1, 3, 2, 4, 5).reshape(B, H, W, c)


def window_attention(xq: torch.Tensor, kv_img: "torch.Tensor | None", P: Params, pfx: str, heads: int) -> torch.Tensor:
    """xq (B, H, W, C); kv_img None (self attention) or (B, H, W, 2C) = cat(k, v) along the channels."""
    B, H, W, C = xq.shape
    hd = C // heads
    xw = _partition(xq)
    q = F.linear(xw, P[pfx + "to_qkv.to_q.weight"], P[pfx + "to_qkv.to_q.bias"])
    if kv_img is None:
        kv = F.linear(xw, P[pfx + "to_qkv.to_kv_from_q.weight"], P[pfx + "to_qkv.to_kv_from_q.bias"])
    else:
        kv = F.linear(_partition(kv_img), P[pfx + "to_qkv.to_kv.weight"], P[pfx + "to_qkv.to_kv.bias"])
    B_ = xw.shape[0]
    q = q.reshape(B_, WS * WS, heads, hd).permute(0, 2, 1, 3)
    kv = kv.reshape(B_, WS * WS, 2, heads, hd).permute(2, 0, 3, 1, 4)
    k, v = kv[0], kv[1]
    attn = (q * hd ** -0.5) @ k.transpose(-2, -1)
    bias = P[pfx + "relative_position_bias_table"][relative_position_index().view(-1)].view(WS * WS, WS * WS, heads).permute(2, 0, 1)
    attn = attn + bias.unsqueeze(0)
    a0, a1 = torch.softmax(attn, dim=-1), torch.relu(attn) ** 2
    w = P[pfx + "w"]
    w1, w2 = torch.exp(w[0]) / torch.sum(torch.exp(w)), torch.exp(w[1]) / torch.sum(torch.exp(w))
    out = ((a0 * w1 + a1 * w2) @ v).transpose(1, 2).reshape(B_, WS * WS, C)
    out = F.linear(out, P[pfx + "proj.weight"], P[pfx + "proj.bias"])
    return _reverse(out, B, H, W)


def _sub(P: Params, pfx: str) -> Params:
    return {k[len(pfx):]: v for k, v in P.items() if k.startswith(pfx)}


def mdassa(x: torch.Tensor, P: Params, pfx: str, heads: int) -> torch.Tensor:
    """x (B, L, C) -> (B, H, W, C)."""
    B, Ltok, C = x.shape
    H = W = int(round(Ltok ** 0.5))
    xn = F.layer_norm(x, (C,), P[pfx + "norm1.weight"], P[pfx + "norm1.bias"], 1e-5)
    img = xn.view(B, H, W, C)
    xs = xn + window_attention(img, None, P, pfx + "attn.", heads).reshape(B, Ltok, C)
    fq = R.fdfp(img, _sub(P, pfx + "fdfp."))
    kv = F.linear(xs, P[pfx + "conv1x1.weight"].reshape(2 * C, C), P[pfx + "conv1x1.bias"]).view(B, H, W, 2 * C)
    return fq + window_attention(fq, kv, P, pfx + "freq_attn.", heads)


def decoder_block(x: torch.Tensor, enc_out: "torch.Tensor | None", P: Params, heads: int, token_mlp: str = "frfn") -> torch.Tensor:
    if enc_out is not None:
        x = torch.cat([x, enc_out], dim=2)
    B, Ltok, Cm = x.shape
    shortcut = x
    m = mdassa(F.layer_norm(x, (Cm,), P["norm1.weight"], P["norm1.bias"], 1e-5), P, "mdassa.", heads).reshape(B, Ltok, Cm)
    y = m + shortcut
    x2 = F.layer_norm(y, (Cm,), P["norm2.weight"], P["norm2.bias"], 1e-5)
    x3 = (R.frfn if token_mlp == "frfn" else R.leff)(x2, _sub(P, "mlp."))
    return F.linear(y + x3, P["mlp_proj.weight"], P["mlp_proj.bias"])


def grads(fn, inputs: "list[torch.Tensor]", P: Params, probe: torch.Tensor):
    """-> (y, [d inputs], {name: grad or absent})."""
    xs = [t.detach().clone().requires_grad_(True) for t in inputs]
    leaves = {n: (v.detach().clone().requires_grad_(True) if v.is_floating_point() else v) for n, v in P.items()}
    y = fn(*xs, leaves)
    (y * probe).sum().backward()
    return y.detach(), [t.grad for t in xs], {n: v.grad for n, v in leaves.items() if v.is_floating_point() and v.grad is not None}


# ---- samplers and projections (src/model/block.py:42-153) ----
def _img(t: torch.Tensor) -> torch.Tensor:
    B, Ltok, C = t.shape
    H = int(round(Ltok ** 0.5))
    return t.transpose(1, 2).reshape(B, C, H, H)


def _tok(img: torch.Tensor) -> torch.Tensor:
    return img.flatten(2).transpose(1, 2)


def downsample(x: torch.Tensor, P: Params, pfx: str = "") -> torch.Tensor:
    """Downsample.forward (:113-122): conv3x3 C -> C/2 (no bias), PixelUnshuffle(2)."""
    return _tok(F.pixel_unshuffle(F.conv2d(_img(x), P[pfx + "body.0.weight"], padding=1), 2))


def upsample(x: torch.Tensor, P: Params, pfx: str = "") -> torch.Tensor:
    """Upsample.forward (:144-153): conv3x3 C -> 2C (no bias), PixelShuffle(2)."""
    return _tok(F.pixel_shuffle(F.conv2d(_img(x), P[pfx + "body.0.weight"], padding=1), 2))


def input_projection(img: torch.Tensor, P: Params, pfx: str = "") -> torch.Tensor:
    """InputProjection.forward (:58-63): conv 3->8, 8->32, 32->C (3x3, bias), LeakyReLU(0.01); NCHW -> tokens."""
    v = img
    for i in range(3):
        v = F.conv2d(v, P[f"{pfx}proj.{i}.weight"], P[f"{pfx}proj.{i}.bias"], padding=1)
    return _tok(F.leaky_relu(v, 0.01))


def output_projection(x: torch.Tensor, P: Params, pfx: str = "") -> torch.Tensor:
    """OutputProjection.forward (:84-91) on the token image reshaped as model.py:435-437 does: tokens -> NCHW [B, 3, H, W]."""
    v = _img(x)
    for i in range(3):
        v = F.conv2d(v, P[f"{pfx}proj.{i}.weight"], P[f"{pfx}proj.{i}.bias"], padding=1)
    return v


def frfn_net(x: torch.Tensor, P: Params) -> torch.Tensor:
    """MyBigFRFNModel.forward (src/model/model.py:596-643) in eval mode (DropPath = identity), with the token image handed to
    output_proj reshaped as model.py:435-437 shows (the shipped forward crashes there) and WITHOUT the bottleneck
    print (:628).  x: (B, 3, S, S)."""
    from . import encoder_ref as E

    def enc(t, pfx):
        return E.encoder_block(t, _sub(P, pfx))

    def dec(t, e, pfx):
        return decoder_block(t, e, _sub(P, pfx), 4)
    y = input_projection(x, P, "input_proj.")
    conv0 = enc(enc(y, "encoder_0."), "encoder_0_1.")
    conv1 = enc(enc(downsample(conv0, P, "downsample_0."), "encoder_1."), "encoder_1_1.")
    conv2 = enc(enc(downsample(conv1, P, "downsample_1."), "encoder_2."), "encoder_2_1.")
    conv3 = enc(enc(downsample(conv2, P, "downsample_2."), "encoder_3."), "encoder_3_1.")
    bott = dec(downsample(conv3, P, "downsample_3."), None, "bottleneck.")
    dec3 = dec(dec(upsample(bott, P, "upsample_3."), conv3, "decoder_3."), None, "decoder_3_1.")
    dec2 = dec(dec(upsample(dec3, P, "upsample_2."), conv2, "decoder_2."), None, "decoder_2_1.")
    dec1 = dec(dec(upsample(dec2, P, "upsample_1."), conv1, "decoder_1."), None, "decoder_1_1.")
    dec0 = dec(dec(upsample(dec1, P, "upsample_0."), conv0, "decoder_0."), None, "decoder_0_1.")
    return output_projection(dec0, P, "output_proj.") + x
