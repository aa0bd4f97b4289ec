"""CPU oracle of the reference's EncoderBlock in its shipped ("Fourier") mode - TEST INFRASTRUCTURE ONLY.

Restates ``EncoderBlock.forward`` (src/model/model.py:57-92) for ``token_mlp = freq_mlp = "frfn"``, ``use_dwt = "Fourier"``,
``drop_path = 0`` (encoder_0 / encoder_0_1 of MyBigFRFNModel, model.py:481-489; DropPath(0) is nn.Identity, :27):

    shortcut = x
    freq_x = norm2(x)                      # computed and then overwritten: norm2 never reaches the output (:62,72)
    x = mlp(norm1(x))                      # FRFN (block.py:249-282)
    freq_x = fft.fftn(x, dim=(-2, -1)).real        # 2-D DFT over the (L, C) axes of the (B, L, C) tokens - not a spatial fft2
    freq_x = freq_mlp(freq_x)              # FRFN
    freq_x = fft.ifftn(freq_x, dim=(-2, -1)).real
    return shortcut + freq_x + x

Parameter names are the reference's: ``norm1.*``, ``mlp.*``, ``norm2.*`` (dead), ``freq_mlp.*`` (registration order :28-52).
Pinned by tests/golden/encoder_ref.npz (the reference's own class, recorded by tests/golden/make_encoder_vectors.py).
"""
from __future__ import annotations

from typing import Dict

import numpy as np
import torch
import torch.nn.functional as F

from .cpu_ref import _hash_uniform, _salt
from . import frfn_ref as R

Params = Dict[str, torch.Tensor]


def param_shapes(dim: int, hidden: int) -> "Dict[str, tuple]":
    s = {"norm1.weight": (dim,), "norm1.bias": (dim,)}
    s.update({"mlp." + k: v for k, v in R.param_shapes(dim, hidden).items()})
    s.update({"norm2.weight": (dim,), "norm2.bias": (dim,)})
    s.update({"freq_mlp." + k: v for k, v in R.param_shapes(dim, hidden).items()})
    return s


def is_dead_param(name: str) -> bool:
    return name.startswith("norm2.")


def closed_form_params(dim: int, hidden: int, dtype=torch.float32) -> Params:
    P = {}
    for n, shp in param_shapes(dim, hidden).items():
        numel = int(np.prod(shp))
        h = _hash_uniform(numel, _salt("enc." + n) + dim).reshape(shp)
        if n.startswith("norm") and n.endswith("weight"):
            v = 1.0 + 0.2 * h
        elif len(shp) == 1:
            v = 0.1 * h
        else:
            v = h * (int(np.prod(shp[1:])) ** -0.5)
        P[n] = torch.from_numpy(np.ascontiguousarray(v)).to(dtype)
    return P


def _sub(P: Params, pfx: str) -> Params:
    return {k[len(pfx):]: v for k, v in P.items() if k.startswith(pfx)}


def encoder_block(x: torch.Tensor, P: Params) -> torch.Tensor:
    C = x.shape[-1]
    x1 = R.frfn(F.layer_norm(x, (C,), P["norm1.weight"], P["norm1.bias"], 1e-5), _sub(P, "mlp."))
    f = torch.fft.fftn(x1, dim=(-2, -1)).real
    f = R.frfn(f, _sub(P, "freq_mlp."))
    f = torch.fft.ifftn(f, dim=(-2, -1)).real
    return x + f + x1


def out_and_grads(x: torch.Tensor, P: Params, probe: torch.Tensor):
    x = x.detach().clone().requires_grad_(True)
    leaves = {n: v.detach().clone().requires_grad_(True) for n, v in P.items()}
    y = encoder_block(x, leaves)
    (y * probe).sum().backward()
    return y.detach(), x.grad, {n: v.grad for n, v in leaves.items() if v.grad is not None}
