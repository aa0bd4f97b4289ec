"""Record fixtures of the reference's EncoderBlock (src/model/model.py:16-93) in its shipped Fourier mode - build container only.

Imported from /root/reference as `src.model.model`.  Module-level imports of model.py / block.py need third-party names that
are absent here and that this block never calls; stand-ins:
  timm.layers.{trunc_normal_, to_2tuple, DropPath}  torch.nn.init.trunc_normal_, a tuple helper, an identity DropPath
                                                    (only constructed for drop_path > 0; the recorded blocks use 0.0)
  pywt                                              empty module (only DWT_2D.__init__ reads it; Fourier mode builds none)
  torchvision.transforms.functional                 empty module (imported at model.py:5, unused by EncoderBlock)
Closed-form weights / inputs (oracle.encoder_ref), fp64 run, stored as float32.

    python tests/golden/make_encoder_vectors.py   ->  tests/golden/encoder_ref.npz
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference")

timm, layers = types.ModuleType("timm"), types.ModuleType("timm.layers")
layers.trunc_normal_ = torch.nn.init.trunc_normal_
layers.to_2tuple = lambda x: x if isinstance(x, tuple) else (x, x)


class _DropPath(torch.nn.Identity):
    def __init__(self, p=0.0):
        super().__init__()


layers.DropPath = _DropPath
timm.layers = layers
tv, tvt, tvf = types.ModuleType("torchvision"), types.ModuleType("torchvision.transforms"), types.ModuleType("torchvision.transforms.functional")
tv.transforms, tvt.functional = tvt, tvf
for name, mod in (("timm", timm), ("timm.layers", layers), ("pywt", types.ModuleType("pywt")), ("torchvision", tv),
                  ("torchvision.transforms", tvt), ("torchvision.transforms.functional", tvf)):
    sys.modules.setdefault(name, mod)

from src.model.model import EncoderBlock  # noqa: E402
from oracle import encoder_ref as E, frfn_ref as R  # noqa: E402

torch.set_num_threads(8)
out = {}
for name, (dim, B, S) in {"d32_b2_s16": (32, 2, 16), "d64_b1_s8": (64, 1, 8)}.items():
    m = EncoderBlock(dim=dim, input_resolution=(S, S), num_heads=4, mlp_ratio=4, token_mlp="frfn", drop_path=0.0,
                     freq_mlp="frfn", use_dwt="Fourier").double()                    # MyBigFRFNModel's encoder_0 (model.py:481-484)
    P = {k: v.double() for k, v in E.closed_form_params(dim, 4 * dim, torch.float32).items()}
    assert list(m.state_dict().keys()) == list(P.keys()), (list(m.state_dict().keys()), list(P.keys()))
    m.load_state_dict(P)
    x = R.closed_form_input(B, S * S, dim, 41, torch.float32).double().requires_grad_(True)
    probe = R.closed_form_input(B, S * S, dim, 43, torch.float32).double()
    y = m(x)
    (y * probe).sum().backward()
    out[name + ".y"] = y.detach().numpy().astype(np.float32)
    out[name + ".dx"] = x.grad.numpy().astype(np.float32)
    dead = []
    for n, p in m.named_parameters():
        if p.grad is None:
            dead.append(n)
        else:
            out[name + ".grad." + n] = p.grad.numpy().astype(np.float32)
    out[name + ".dead"] = np.array(dead)
    print(name, "y", float(y.detach().abs().max()), "no-grad params:", dead)
out["names"] = np.array(list(E.param_shapes(32, 128).keys()))
np.savez_compressed(os.path.join(HERE, "encoder_ref.npz"), **out)
print("wrote encoder_ref.npz", os.path.getsize(os.path.join(HERE, "encoder_ref.npz")) // 1024, "KB")
