"""Record fixtures of the reference's MDASSA (src/model/block.py:373-515) and DecoderBlock (src/model/model.py:96-160) -
build container only.  Same stand-ins as make_encoder_vectors.py / make_frfn_vectors.py for module-level third-party imports
(timm.layers helpers, torchvision.transforms.functional, PyWavelets' published Haar filter bank for the DWT buffers FDFP
constructs).  Closed-form parameters by name / shape (oracle.decoder_ref.closed_form_params), fp64 run, stored as float32.

    python tests/golden/make_decoder_vectors.py   ->  tests/golden/decoder_ref.npz
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
pywt = types.ModuleType("pywt")


class _Haar:
    dec_lo = [0.7071067811865476, 0.7071067811865476]
    dec_hi = [-0.7071067811865476, 0.7071067811865476]
    rec_lo = [0.7071067811865476, 0.7071067811865476]
    rec_hi = [0.7071067811865476, -0.7071067811865476]

    def __init__(self, name):
        assert name == "haar"


pywt.Wavelet = _Haar
for name, mod in (("timm", timm), ("timm.layers", layers), ("pywt", pywt), ("torchvision", tv), ("torchvision.transforms", tvt),
                  ("torchvision.transforms.functional", tvf)):
    sys.modules.setdefault(name, mod)

from src.model.block import MDASSA, Downsample, Upsample, InputProjection, OutputProjection  # noqa: E402
from src.model.model import DecoderBlock  # noqa: E402
from oracle import decoder_ref as D, frfn_ref as R  # noqa: E402

torch.set_num_threads(8)
out = {}


def record(tag, module, inputs, probe, salt):
    module = module.double()
    sd = module.state_dict()
    shapes = {k: tuple(v.shape) for k, v in sd.items() if v.is_floating_point() and not k.split(".")[-2:][0] in ("dwt", "idwt")
              and ".dwt." not in k and ".idwt." not in k and not k.startswith(("dwt.", "idwt."))}
    P = {k: v.double() for k, v in D.closed_form_params(shapes, salt, torch.float32).items()}
    module.load_state_dict(P, strict=False)
    xs = [t.double().requires_grad_(True) for t in inputs]
    y = module(*xs)
    (y * probe.double().reshape(y.shape)).sum().backward()
    out[tag + ".keys"] = np.array(list(sd.keys()))
    out[tag + ".param_names"] = np.array(list(shapes.keys()))
    out[tag + ".param_shapes"] = np.array([str(shapes[k]) for k in shapes])
    out[tag + ".y"] = y.detach().numpy().astype(np.float32)
    for i, t in enumerate(xs):
        out[f"{tag}.dx{i}"] = t.grad.numpy().astype(np.float32)
    dead = []
    for n, p in module.named_parameters():
        if p.grad is None:
            dead.append(n)
        else:
            out[tag + ".grad." + n] = p.grad.numpy().astype(np.float32)
    out[tag + ".dead"] = np.array(dead)
    print(tag, "y", tuple(y.shape), float(y.detach().abs().max()), "no-grad:", dead)


S = 16
x64 = R.closed_form_input(1, S * S, 64, 71)
pr64 = R.closed_form_input(1, S * S, 64, 72)
record("mdassa_d64", MDASSA(64, win_size=8, shift_size=0, num_heads=4, token_projection="linear", qkv_bias=True, qk_scale=None,
                            attn_drop=0., proj_drop=0., enc_out=True, freq_attn_win_ratio=2, use_dwt="Fourier"), [x64], pr64, "mdassa64")
x32, e32, pr32 = R.closed_form_input(1, S * S, 32, 73), R.closed_form_input(1, S * S, 32, 74), R.closed_form_input(1, S * S, 32, 75)
record("dec_d32_enc", DecoderBlock(dim=32, input_resolution=(S, S), num_heads=4, win_size=8, shift_size=0, mlp_ratio=4, token_mlp="frfn",
                                   drop_path=0.0, token_projection="linear", enc_out=True, freq_attn_win_ratio=8, use_dwt="Fourier"),
       [x32, e32], pr32, "dec32")
record("dec_d64_noenc", DecoderBlock(dim=64, input_resolution=(S, S), num_heads=4, win_size=8, shift_size=0, mlp_ratio=4, token_mlp="frfn",
                                     drop_path=0.0, token_projection="linear", enc_out=False, freq_attn_win_ratio=2, use_dwt="Fourier"),
       [x64], pr64, "dec64")
# samplers and projections (block.py:42-153).  OutputProjection takes NCHW in the reference: it is fed the token image reshaped
# as model.py:435-437 does (tokens -> (B, C, H, W)); the recorded input gradient is w.r.t. the tokens.
record("down_c32", Downsample(32, 64), [x32], R.closed_form_input(1, (S // 2) ** 2, 64, 76), "down32")
record("up_c64", Upsample(64, 32), [R.closed_form_input(1, (S // 2) ** 2, 64, 77)], R.closed_form_input(1, S * S, 32, 78), "up64")
img = R.closed_form_input(1, 3, S * S, 79).reshape(1, 3, S, S)
record("inproj_c32", InputProjection(in_channels=3, out_channels=32), [img], pr32, "inproj32")


class _OutOnTokens(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.inner = OutputProjection(in_channels=32, out_channel=3, kernel_size=3, stride=1, norm_layer=None, act_layer=None)

    def forward(self, t):
        B, Ltok, C = t.shape
        return self.inner(t.transpose(1, 2).reshape(B, C, S, S))


record("outproj_c32", _OutOnTokens(), [x32], R.closed_form_input(1, 3, S * S, 80).reshape(1, 3, S, S), "outproj32")
np.savez_compressed(os.path.join(HERE, "decoder_ref.npz"), **out)
print("wrote decoder_ref.npz", os.path.getsize(os.path.join(HERE, "decoder_ref.npz")) // 1024, "KB")
