"""CPU oracle of the reference's FRFN feed-forward - TEST INFRASTRUCTURE ONLY (same rule as cpu_ref.py).

Functional restatement of ``FRFN.forward`` (src/model/block.py:262-282) on the reference's parameter names
(``linear1.0.{weight,bias}`` [8C? no: 2*hidden, C], ``dwconv.0.{weight,bias}`` [hidden,1,3,3], ``linear2.0.{weight,bias}``
[C, hidden], ``partial_conv3.weight`` [C/4, C/4, 3, 3]; ``__init__`` :250-260):

    x: (B, L, C), L = H*W square   ->  image (B, C, H, W)                                   :263-266
    x1, x2 = split(C/4, 3C/4); x1 = partial_conv3(x1) (3x3, pad 1, no bias); cat            :268-270
    x = GELU(linear1(x))  (B, L, 2*hidden); x_1, x_2 = chunk(2)                             :274-275
    x_1 = GELU(dwconv3x3(x_1) + bias) as an image                                           :277-278
    x = linear2(x_1 * x_2)                                                                  :280-282

Pinned by tests/golden/frfn_ref.npz (outputs and gradients of the reference's own class, recorded in fp64 by
tests/golden/make_frfn_vectors.py).  Nothing under underwater_image_restoration_amd/ imports this file.
"""
from __future__ import annotations

import math
from typing import Dict

import numpy as np
import torch
import torch.nn.functional as F

from .cpu_ref import _hash_uniform, _salt

Params = Dict[str, torch.Tensor]


def param_shapes(dim: int, hidden: int) -> "Dict[str, tuple]":
    """Registration order of the reference's __init__ (block.py:250-260) = state_dict order."""
    c4 = dim // 4
    return {"linear1.0.weight": (2 * hidden, dim), "linear1.0.bias": (2 * hidden,),
            "dwconv.0.weight": (hidden, 1, 3, 3), "dwconv.0.bias": (hidden,),
            "linear2.0.weight": (dim, hidden), "linear2.0.bias": (dim,),
            "partial_conv3.weight": (c4, c4, 3, 3)}


def closed_form_params(dim: int, hidden: int, dtype=torch.float32) -> Params:
    """Deterministic weights without any RNG: hash-uniform(-1, 1) * fan_in^-1/2 (biases * 0.1)."""
    P = {}
    for n, shp in param_shapes(dim, hidden).items():
        numel = int(np.prod(shp))
        fan_in = int(np.prod(shp[1:])) if len(shp) > 1 else 1
        scale = fan_in ** -0.5 if len(shp) > 1 else 0.1
        P[n] = torch.from_numpy(_hash_uniform(numel, _salt("frfn." + n) + dim).reshape(shp) * scale).to(dtype)
    return P


def leff_param_shapes(dim: int, hidden: int) -> "Dict[str, tuple]":
    """LeFF.__init__ (block.py:224-232)."""
    return {"linear1.0.weight": (hidden, dim), "linear1.0.bias": (hidden,),
            "dwconv.0.weight": (hidden, 1, 3, 3), "dwconv.0.bias": (hidden,),
            "linear2.0.weight": (dim, hidden), "linear2.0.bias": (dim,)}


def closed_form_leff_params(dim: int, hidden: int, dtype=torch.float32) -> Params:
    P = {}
    for n, shp in leff_param_shapes(dim, hidden).items():
        numel = int(np.prod(shp))
        fan_in = int(np.prod(shp[1:])) if len(shp) > 1 else 1
        scale = fan_in ** -0.5 if len(shp) > 1 else 0.1
        P[n] = torch.from_numpy(_hash_uniform(numel, _salt("leff." + n) + dim).reshape(shp) * scale).to(dtype)
    return P


def leff(x: torch.Tensor, P: Params) -> torch.Tensor:
    """LeFF.forward (block.py:234-247): Linear + GELU -> depthwise 3x3 (+ bias) + GELU on the token image -> Linear."""
    B, L, C = x.shape
    H = W = int(math.sqrt(L))
    hidden = P["dwconv.0.weight"].shape[0]
    t = F.gelu(F.linear(x, P["linear1.0.weight"], P["linear1.0.bias"]))
    u = F.conv2d(t.reshape(B, H, W, hidden).permute(0, 3, 1, 2), P["dwconv.0.weight"], P["dwconv.0.bias"], padding=1, groups=hidden)
    g = F.gelu(u).permute(0, 2, 3, 1).reshape(B, L, hidden)
    return F.linear(g, P["linear2.0.weight"], P["linear2.0.bias"])


def leff_out_and_grads(x: torch.Tensor, P: Params, probe: torch.Tensor):
    x = x.detach().clone().requires_grad_(True)
    leaves = {n: v.detach().clone().requires_grad_(True) for n, v in P.items()}
    y = leff(x, leaves)
    (y * probe).sum().backward()
    return y.detach(), x.grad, {n: v.grad for n, v in leaves.items()}


def fdfp_param_shapes(dim: int, hidden: int) -> "Dict[str, tuple]":
    """FDFP.__init__ with use_dwt="Fourier" (block.py:519-530): no DWT sub-modules, two 1x1 convolutions with bias."""
    return {"conv1.weight": (hidden, dim, 1, 1), "conv1.bias": (hidden,), "conv2.weight": (dim, hidden, 1, 1), "conv2.bias": (dim,)}


def closed_form_fdfp_params(dim: int, hidden: int, dtype=torch.float32) -> Params:
    P = {}
    for n, shp in fdfp_param_shapes(dim, hidden).items():
        numel = int(np.prod(shp))
        scale = int(np.prod(shp[1:])) ** -0.5 if len(shp) > 1 else 0.1
        P[n] = torch.from_numpy(_hash_uniform(numel, _salt("fdfp." + n) + dim).reshape(shp) * scale).to(dtype)
    return P


def fdfp(x: torch.Tensor, P: Params) -> torch.Tensor:
    """FDFP.forward, Fourier mode (block.py:532-554): x (B, H, W, C)."""
    v = x.permute(0, 3, 1, 2)
    v = torch.fft.fftn(v, dim=(-2, -1)).real
    v = F.conv2d(F.gelu(F.conv2d(v, P["conv1.weight"], P["conv1.bias"])), P["conv2.weight"], P["conv2.bias"])
    v = torch.fft.ifftn(v, dim=(-2, -1)).real
    return v.permute(0, 2, 3, 1)


def fdfp_out_and_grads(x: torch.Tensor, P: Params, probe: torch.Tensor):
    x = x.detach().clone().requires_grad_(True)
    leaves = {n: v.detach().clone().requires_grad_(True) for n, v in P.items()}
    y = fdfp(x, leaves)
    (y * probe).sum().backward()
    return y.detach(), x.grad, {n: v.grad for n, v in leaves.items()}


def closed_form_input(b: int, l: int, c: int, salt: int, dtype=torch.float32) -> torch.Tensor:
    return torch.from_numpy(_hash_uniform(b * l * c, 7000 + salt).reshape(b, l, c)).to(dtype)


def frfn(x: torch.Tensor, P: Params, hw: "tuple | None" = None) -> torch.Tensor:
    B, L, C = x.shape
    H = W = int(math.sqrt(L)) if hw is None else 0
    if hw is not None:
        H, W = hw
    hidden = P["dwconv.0.weight"].shape[0]
    c4 = C // 4
    img = x.reshape(B, H, W, C).permute(0, 3, 1, 2)
    x1 = F.conv2d(img[:, :c4], P["partial_conv3.weight"], padding=1)
    img = torch.cat((x1, img[:, c4:]), dim=1)
    t = F.gelu(F.linear(img.permute(0, 2, 3, 1).reshape(B, L, C), P["linear1.0.weight"], P["linear1.0.bias"]))
    x_1, x_2 = t.chunk(2, dim=-1)
    u = F.conv2d(x_1.reshape(B, H, W, hidden).permute(0, 3, 1, 2), P["dwconv.0.weight"], P["dwconv.0.bias"], padding=1,
                 groups=hidden)
    x_1 = F.gelu(u).permute(0, 2, 3, 1).reshape(B, L, hidden)
    return F.linear(x_1 * x_2, P["linear2.0.weight"], P["linear2.0.bias"])


def out_and_grads(x: torch.Tensor, P: Params, probe: torch.Tensor, hw=None):
    """-> (y, dx, {name: grad}) for the scalar sum(y * probe)."""
    x = x.detach().clone().requires_grad_(True)
    leaves = {n: v.detach().clone().requires_grad_(True) for n, v in P.items()}
    y = frfn(x, leaves, hw)
    (y * probe).sum().backward()
    return y.detach(), x.grad, {n: v.grad for n, v in leaves.items()}
