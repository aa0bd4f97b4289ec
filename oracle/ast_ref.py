"""CPU oracle of the reference's ``--arch AST`` network - TEST INFRASTRUCTURE ONLY.

Functional restatement, on the reference's ``state_dict`` names, of src/Models/AST.py as ``AST()`` builds it with its defaults
(embed_dim 32, depths 2 x 9, win_size 8, token_mlp 'leff', shift_flag True; encoders att=False, bottleneck / decoders
att=True + sparseAtt=True):
  InputProj.forward            AST.py:461-466     conv3x3 3 -> C + bias, LeakyReLU(0.01), image -> tokens
  OutputProj.forward           :485-493           tokens -> image, conv3x3 C -> 3 + bias
  Downsample.forward           :417-424           Conv2d(k 4, s 2, p 1)
  Upsample.forward             :437-443           ConvTranspose2d(k 2, s 2)
  LeFF.forward                 :307-326           (same arithmetic as src/model/block.py's LeFF: oracle.frfn_ref.leff)
  LinearProjection.forward     :52-64             to_q, to_kv on the same window
  WindowAttention_sparse       :187-219           q scale, + relative position bias, (+ mask), softmax * w1 + relu^2 * w2, proj
  TransformerBlock.forward     :552-623           shift mask (:568-589), cyclic shift, window partition / reverse, two residuals,
                                                  the per-sample stochastic-depth factors given explicitly (``dp``)
  BasicASTLayer.forward        :671-677
  AST.forward                  :885-921
Pinned by tests/golden/ast_ref.npz (outputs and gradients of the reference's own classes, tests/golden/make_ast_vectors.py).
"""
from __future__ import annotations

from typing import Dict

import torch
import torch.nn.functional as F

from . import decoder_ref as D, frfn_ref as R

Params = Dict[str, torch.Tensor]
WS = 8
closed_form_params = D.closed_form_params
closed_form_input = R.closed_form_input
grads = D.grads

# AST() defaults (AST.py:681-687)
EMBED = 32
HEADS = (1, 2, 4, 8, 16, 16, 8, 4, 2)
LAYERS = ("encoderlayer_0", "encoderlayer_1", "encoderlayer_2", "encoderlayer_3", "conv", "decoderlayer_0", "decoderlayer_1",
          "decoderlayer_2", "decoderlayer_3")
LAYER_DIM = (1, 2, 4, 8, 16, 16, 8, 4, 2)          # x embed_dim
LAYER_ATT = (False, False, False, False, True, True, True, True, True)


def drop_path_rates(drop_path_rate: float = 0.1):
    """Per-layer [block 0, block 1] stochastic-depth probabilities (AST.py:705-707 and the slices of :725-845)."""
    enc = [x.item() for x in torch.linspace(0, drop_path_rate, 8)]
    dec = enc[::-1]
    return [enc[0:2], enc[2:4], enc[4:6], enc[6:8], [drop_path_rate] * 2, dec[0:2], dec[2:4], dec[4:6], dec[6:8]]


def shift_mask(H: int, W: int, shift: int, dtype=torch.float32) -> torch.Tensor:
    """[nW, 64, 64] additive mask of the shifted windows (AST.py:568-589): 0 inside a region, -100 across regions."""
    m = torch.zeros((1, H, W, 1), dtype=dtype)
    cnt = 0
    for h in (slice(0, -WS), slice(-WS, -shift), slice(-shift, None)):
        for w in (slice(0, -WS), slice(-WS, -shift), slice(-shift, None)):
            m[:, h, w, :] = cnt
            cnt += 1
    mw = D._partition(m).view(-1, WS * WS)
    am = mw.unsqueeze(1) - mw.unsqueeze(2)
    return am.masked_fill(am != 0, -100.0).masked_fill(am == 0, 0.0)


def window_attention(xw: torch.Tensor, P: Params, pfx: str, heads: int, mask: "torch.Tensor | None") -> torch.Tensor:
    """WindowAttention_sparse.forward on windows xw [B_, 64, C]; mask [nW, 64, 64] or None."""
    B_, N, C = xw.shape
    hd = C // heads
    q = F.linear(xw, P[pfx + "qkv.to_q.weight"], P[pfx + "qkv.to_q.bias"]).reshape(B_, N, heads, hd).permute(0, 2, 1, 3)
    kv = F.linear(xw, P[pfx + "qkv.to_kv.weight"], P[pfx + "qkv.to_kv.bias"]).reshape(B_, N, 2, heads, hd).permute(2, 0, 3, 1, 4)
    k, v = kv[0], kv[1]
    attn = (q * hd ** -0.5) @ k.transpose(-2, -1)
    bias = P[pfx + "relative_position_bias_table"][D.relative_position_index().view(-1)].view(N, N, heads).permute(2, 0, 1)
    attn = attn + bias.unsqueeze(0)
    if mask is not None:
        nW = mask.shape[0]
        attn = (attn.view(B_ // nW, nW, heads, N, N) + mask.to(attn.dtype).unsqueeze(1).unsqueeze(0)).view(-1, heads, N, N)
    a0, a1 = torch.softmax(attn, dim=-1), torch.relu(attn) ** 2
    w = P[pfx + "w"]
    w1, w2 = torch.exp(w[0]) / torch.sum(torch.exp(w)), torch.exp(w[1]) / torch.sum(torch.exp(w))
    out = ((a0 * w1 + a1 * w2) @ v).transpose(1, 2).reshape(B_, N, C)
    return F.linear(out, P[pfx + "proj.weight"], P[pfx + "proj.bias"])


def transformer_block(x: torch.Tensor, P: Params, pfx: str, heads: int, shift: int, att: bool, dp=None) -> torch.Tensor:
    """x (B, L, C).  dp: None (eval / drop_path 0) or a pair of per-sample factor tensors [B] (attention branch, mlp branch) -
    what timm's DropPath multiplies each residual branch with in training mode."""
    B, Ltok, C = x.shape
    H = W = int(round(Ltok ** 0.5))
    f = (lambda i: 1.0) if dp is None else (lambda i: dp[i].to(x.dtype).view(B, 1, 1))
    if att:
        xn = F.layer_norm(x, (C,), P[pfx + "norm1.weight"], P[pfx + "norm1.bias"], 1e-5).view(B, H, W, C)
        mask = None
        if shift > 0:
            xn = torch.roll(xn, shifts=(-shift, -shift), dims=(1, 2))
            mask = shift_mask(H, W, shift, x.dtype)
        aw = window_attention(D._partition(xn), P, pfx + "attn.", heads, mask)
        a = D._reverse(aw, B, H, W)
        if shift > 0:
            a = torch.roll(a, shifts=(shift, shift), dims=(1, 2))
        x = x + f(0) * a.reshape(B, Ltok, C)
    m = R.leff(F.layer_norm(x, (C,), P[pfx + "norm2.weight"], P[pfx + "norm2.bias"], 1e-5), D._sub(P, pfx + "mlp."))
    return x + f(1) * m


def input_proj(img: torch.Tensor, P: Params, pfx: str = "") -> torch.Tensor:
    return D._tok(F.leaky_relu(F.conv2d(img, P[pfx + "proj.0.weight"], P[pfx + "proj.0.bias"], padding=1), 0.01))


def output_proj(x: torch.Tensor, P: Params, pfx: str = "") -> torch.Tensor:
    return F.conv2d(D._img(x), P[pfx + "proj.0.weight"], P[pfx + "proj.0.bias"], padding=1)


def downsample(x: torch.Tensor, P: Params, pfx: str = "") -> torch.Tensor:
    return D._tok(F.conv2d(D._img(x), P[pfx + "conv.0.weight"], P[pfx + "conv.0.bias"], stride=2, padding=1))


def upsample(x: torch.Tensor, P: Params, pfx: str = "") -> torch.Tensor:
    return D._tok(F.conv_transpose2d(D._img(x), P[pfx + "deconv.0.weight"], P[pfx + "deconv.0.bias"], stride=2))


def layer(x: torch.Tensor, P: Params, idx: int, dp=None) -> torch.Tensor:
    """BasicASTLayer number idx of LAYERS (depth 2: block 0 un-shifted, block 1 shifted by win_size // 2).  dp: None or
    {(block, branch): factors [B]}."""
    pfx = LAYERS[idx] + ".blocks."
    for b in (0, 1):
        d = None if dp is None else (dp.get((idx, b, 0)), dp.get((idx, b, 1)))
        if d is not None:
            one = torch.ones(x.shape[0], dtype=x.dtype)
            d = tuple(one if t is None else t for t in d)
        x = transformer_block(x, P, f"{pfx}{b}.", HEADS[idx], WS // 2 if b == 1 else 0, LAYER_ATT[idx], d)
    return x


def ast_net(x: torch.Tensor, P: Params, dp=None) -> torch.Tensor:
    """AST.forward (AST.py:885-921), mask None, dd_in 3.  x (B, 3, S, S), S a multiple of 128."""
    y = input_proj(x, P, "input_proj.")
    conv0 = layer(y, P, 0, dp)
    conv1 = layer(downsample(conv0, P, "dowsample_0."), P, 1, dp)
    conv2 = layer(downsample(conv1, P, "dowsample_1."), P, 2, dp)
    conv3 = layer(downsample(conv2, P, "dowsample_2."), P, 3, dp)
    conv4 = layer(downsample(conv3, P, "dowsample_3."), P, 4, dp)
    d0 = layer(torch.cat([upsample(conv4, P, "upsample_0."), conv3], -1), P, 5, dp)
    d1 = layer(torch.cat([upsample(d0, P, "upsample_1."), conv2], -1), P, 6, dp)
    d2 = layer(torch.cat([upsample(d1, P, "upsample_2."), conv1], -1), P, 7, dp)
    d3 = layer(torch.cat([upsample(d2, P, "upsample_3."), conv0], -1), P, 8, dp)
    return x + output_proj(d3, P, "output_proj.")
