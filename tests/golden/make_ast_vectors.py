"""Record fixtures of the reference's ``AST`` network and its building blocks (src/Models/AST.py) - build container only.
AST.py is loaded BY FILE PATH (``import src.Models`` would pull torchvision / pywt); its module-level third-party import
``from timm.layers import DropPath, to_2tuple, trunc_normal_`` is served by tests/golden/_ref_stubs.py (torch's own
trunc_normal_, a tuple helper, timm's published drop_path function) - timm is not in the image.  Closed-form parameters by
name / shape (oracle.ast_ref.closed_form_params), fp64 runs, stored as float32.

    python tests/golden/make_ast_vectors.py   ->  tests/golden/ast_ref.npz, tests/golden/ast_manifest.npz
"""
import importlib.util
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
import _ref_stubs  # noqa: E402

_ref_stubs.install()
spec = importlib.util.spec_from_file_location("ref_ast", "/root/reference/src/Models/AST.py")
A = importlib.util.module_from_spec(spec)
spec.loader.exec_module(A)
from oracle import ast_ref as O  # noqa: E402

torch.set_num_threads(8)
out = {}


def record(tag, module, inputs, probe, salt, full_grads=True, keep=()):
    module = module.double().eval()
    sd = module.state_dict()
    shapes = {k: tuple(v.shape) for k, v in sd.items() if v.is_floating_point()}
    P = {k: v.double() for k, v in O.closed_form_params(shapes, salt, torch.float32).items()}
    module.load_state_dict(P, strict=False)
    xs = [t.double().requires_grad_(True) for t in inputs]
    y = module(*xs)
    (y * probe.double().reshape(y.shape)).sum().backward()
    out[tag + ".keys"] = np.array(list(sd.keys()))
    out[tag + ".param_names"] = np.array(list(shapes.keys()))
    out[tag + ".param_shapes"] = np.array([str(shapes[k]) for k in shapes])
    out[tag + ".y"] = y.detach().numpy().astype(np.float32)
    for i, t in enumerate(xs):
        if t.grad is not None:
            out[f"{tag}.dx{i}"] = t.grad.numpy().astype(np.float32)
    dead, names, norms = [], [], []
    for n, p in module.named_parameters():
        if p.grad is None:
            dead.append(n)
            continue
        names.append(n)
        norms.append(float(p.grad.norm()))
        if full_grads or n in keep:
            out[tag + ".grad." + n] = p.grad.numpy().astype(np.float32)
    out[tag + ".dead"] = np.array(dead)
    out[tag + ".grad_names"] = np.array(names)
    out[tag + ".grad_norms"] = np.array(norms, dtype=np.float64)
    print(tag, "y", tuple(y.shape), float(y.detach().abs().max()), "no-grad:", dead)


S = 16


def block(dim, heads, shift, att):
    return A.TransformerBlock(dim=dim, input_resolution=(S, S), num_heads=heads, win_size=8, shift_size=shift, mlp_ratio=4.,
                              qkv_bias=True, qk_scale=None, drop=0., attn_drop=0., drop_path=0.05, token_projection="linear",
                              token_mlp="leff", att=att, sparseAtt=att)


x64, pr64 = O.closed_form_input(2, S * S, 64, 171), O.closed_form_input(2, S * S, 64, 172)
x32, pr32 = O.closed_form_input(2, S * S, 32, 173), O.closed_form_input(2, S * S, 32, 174)
record("blk_att_s0", block(64, 2, 0, True), [x64], pr64, "astblk0")
record("blk_att_s4", block(64, 2, 4, True), [x64], pr64, "astblk4")
record("blk_noatt", block(32, 1, 0, False), [x32], pr32, "astblkn")
record("down_32", A.Downsample(32, 64), [x32], O.closed_form_input(2, (S // 2) ** 2, 64, 175), "astdown")
record("up_64", A.Upsample(64, 32), [x64], O.closed_form_input(2, 4 * S * S, 32, 176), "astup")
img = O.closed_form_input(2, 3, S * S, 177).reshape(2, 3, S, S)
record("inproj", A.InputProj(in_channel=3, out_channel=32, kernel_size=3, stride=1, act_layer=torch.nn.LeakyReLU), [img], pr32, "astin")
record("outproj", A.OutputProj(in_channel=64, out_channel=3, kernel_size=3, stride=1), [x64],
       O.closed_form_input(2, 3, S * S, 178).reshape(2, 3, S, S), "astout")

# whole network, 1 x 3 x 128 x 128 (bottleneck = one 8 x 8 window, still cyclically shifted in its second block)
N = 128
ximg = O.closed_form_input(1, 3, N * N, 179).reshape(1, 3, N, N)
net = A.AST()
KEEP = ("input_proj.proj.0.weight", "output_proj.proj.0.bias", "encoderlayer_0.blocks.1.mlp.dwconv.0.weight",
        "dowsample_0.conv.0.weight", "upsample_3.deconv.0.weight", "upsample_0.deconv.0.bias", "conv.blocks.1.attn.relative_position_bias_table",
        "conv.blocks.1.attn.w", "decoderlayer_3.blocks.1.attn.qkv.to_q.weight", "decoderlayer_3.blocks.1.attn.relative_position_bias_table",
        "decoderlayer_1.blocks.0.norm1.weight", "decoderlayer_2.blocks.1.attn.w", "encoderlayer_2.blocks.0.norm2.bias")
record("net128", net, [ximg], O.closed_form_input(1, 3, N * N, 180).reshape(1, 3, N, N), "astnet", full_grads=False, keep=KEEP)
np.savez_compressed(os.path.join(HERE, "ast_ref.npz"), **out)
print("wrote ast_ref.npz", os.path.getsize(os.path.join(HERE, "ast_ref.npz")) // 1024, "KB")

# manifest: state_dict keys / shapes of AST() and its default initialisation under a seed (RNG-stream parity of the factory)
torch.manual_seed(1234)
net = A.AST()
sd = net.state_dict()
man = {"keys": np.array(list(sd.keys())), "shapes": np.array([str(tuple(v.shape)) for v in sd.values()]),
       "sums": np.array([float(v.double().sum()) for v in sd.values()]),
       "abs_sums": np.array([float(v.double().abs().sum()) for v in sd.values()]),
       "n_params": np.array(sum(p.numel() for p in net.parameters())),
       "drop_probs": np.array([float(getattr(b.drop_path, "drop_prob", 0.0)) for l in O.LAYERS for b in getattr(net, l).blocks])}
np.savez_compressed(os.path.join(HERE, "ast_manifest.npz"), **man)
print("wrote ast_manifest.npz", len(sd), "tensors", int(man["n_params"]), "parameters; drop probs", man["drop_probs"])
