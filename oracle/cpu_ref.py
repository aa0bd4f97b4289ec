"""CPU oracle for the SpectralTransformer train-step hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``underwater_image_restoration_amd/``
may import this file; only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` use it, and only as the checker.

This is a functional (stateless) restatement of the *live* path of the
reference, written against a flat ``{name: tensor}`` parameter dict that uses
the reference's own ``state_dict`` key names.  It runs in whatever dtype the
parameters/inputs carry (fp32 = the reference's arithmetic; fp64 = a tighter
yard-stick for the HIP kernels).

Parity pin: ``tests/golden/make_golden.py`` imports the reference
``src/Models/SpectralTransformer.py`` *by file path* in the build container,
loads the closed-form weights of :func:`closed_form_state` into it and stores
its outputs / losses / gradient norms / 2-epoch Adam trajectory under
``tests/golden/*.npz``; ``tests/test_oracle_golden.py`` checks this file
against those vectors.  Citations below are ``path:line`` into the reference.
"""
from __future__ import annotations

import math
from typing import Dict, List, Tuple

import numpy as np
import torch
import torch.nn.functional as F

Params = Dict[str, torch.Tensor]

# ---------------------------------------------------------------------------
# Architecture constants (src/Models/SpectralTransformer.py:214-215, 220-252)
# ---------------------------------------------------------------------------
NUM_BLOCKS = (2, 3, 3, 4)
NUM_HEADS = (1, 2, 4, 8)
CHANNELS = (16, 32, 64, 128)
NUM_REFINEMENT = 4
EXPANSION = 2.66


def hidden_of(c: int) -> int:
    """GDFN hidden width, ``int(channels * 2.66)`` (SpectralTransformer.py:121)."""
    return int(c * EXPANSION)


def stage_table() -> List[Tuple[str, int, int, int]]:
    """(prefix, n_blocks, channels, heads) of every TransformerBlock stack, in
    forward order (SpectralTransformer.py:220-223, 237-246)."""
    t = [(f"encoders.{i}", NUM_BLOCKS[i], CHANNELS[i], NUM_HEADS[i]) for i in range(4)]
    t.append(("decoders.0", NUM_BLOCKS[2], CHANNELS[2], NUM_HEADS[2]))
    t.append(("decoders.1", NUM_BLOCKS[1], CHANNELS[1], NUM_HEADS[1]))
    t.append(("decoders.2", NUM_BLOCKS[0], CHANNELS[1], NUM_HEADS[0]))
    t.append(("refinement", NUM_REFINEMENT, CHANNELS[1], NUM_HEADS[0]))
    return t


# ---------------------------------------------------------------------------
# Parameter inventory: the reference's 443 state_dict tensors, name -> shape
# ---------------------------------------------------------------------------
def _block_shapes(prefix: str, c: int, heads: int) -> List[Tuple[str, Tuple[int, ...]]]:
    h = hidden_of(c)
    return [
        (f"{prefix}.norm1.weight", (c,)), (f"{prefix}.norm1.bias", (c,)),
        (f"{prefix}.attn.temperature", (1, heads, 1, 1)),
        (f"{prefix}.attn.qkv.weight", (3 * c, c, 1, 1)),
        (f"{prefix}.attn.qkv_conv.weight", (3 * c, 1, 3, 3)),
        (f"{prefix}.attn.project_out.weight", (c, c, 1, 1)),
        (f"{prefix}.attn.kv.weight", (2 * c, c, 1, 1)),
        (f"{prefix}.attn.q1X1_1.weight", (c, c, 1, 1)),
        (f"{prefix}.attn.q1X1_2.weight", (c, c, 1, 1)),
        (f"{prefix}.attn.kv_conv.weight", (2 * c, 1, 3, 3)),
        (f"{prefix}.attn.project_outf.weight", (c, c, 1, 1)),
        (f"{prefix}.norm2.weight", (c,)), (f"{prefix}.norm2.bias", (c,)),
        (f"{prefix}.ffn.project_in.weight", (2 * h, c, 1, 1)),
        (f"{prefix}.ffn.conv.weight", (2 * h, 1, 3, 3)),
        (f"{prefix}.ffn.project_out.weight", (c, h, 1, 1)),
    ]


def _ups_shapes(prefix: str, c: int) -> List[Tuple[str, Tuple[int, ...]]]:
    out = []
    for br in ("amp_fuse", "pha_fuse"):
        for i in (0, 2):
            out += [(f"{prefix}.Fups.{br}.{i}.weight", (c, c, 1, 1)), (f"{prefix}.Fups.{br}.{i}.bias", (c,))]
    out += [(f"{prefix}.Fups.post.weight", (c // 2, c, 1, 1)), (f"{prefix}.Fups.post.bias", (c // 2,))]
    out += [(f"{prefix}.Sups.body.0.weight", (2 * c, c, 3, 3))]
    out += [(f"{prefix}.reduce.weight", (c // 2, c, 1, 1))]
    return out


def param_shapes() -> "Dict[str, Tuple[int, ...]]":
    """Ordered name->shape map equal to ``SpectralTransformer().state_dict()``
    (module registration order of SpectralTransformer.py:216-252)."""
    items: List[Tuple[str, Tuple[int, ...]]] = [("embed_conv_rgb.weight", (16, 3, 3, 3))]
    for i in range(4):
        for j in range(NUM_BLOCKS[i]):
            items += _block_shapes(f"encoders.{i}.{j}", CHANNELS[i], NUM_HEADS[i])
    for k, c in (("down1", 16), ("down2", 32), ("down3", 64)):
        items.append((f"{k}.body.0.weight", (c // 2, c, 3, 3)))
    for k, c in (("ups_1", 128), ("ups_2", 64), ("ups_3", 32), ("ups_4", 3)):
        items += _ups_shapes(k, c)
    items.append(("ups1.body.0.weight", (64, 32, 3, 3)))
    items.append(("reduces2.weight", (32, 64, 1, 1)))
    items.append(("reduces1.weight", (64, 128, 1, 1)))
    for pfx, nb, c, hd in stage_table()[4:7]:
        for j in range(nb):
            items += _block_shapes(f"{pfx}.{j}", c, hd)
    for j in range(NUM_REFINEMENT):
        items += _block_shapes(f"refinement.{j}", 32, 1)
    items.append(("output.weight", (3, 8, 3, 3)))
    items.append(("output1.weight", (8, 16, 3, 3)))
    items.append(("ups2.body.0.weight", (32, 16, 3, 3)))
    items.append(("outputl.weight", (8, 32, 3, 3)))
    return dict(items)


def is_dead_param(name: str) -> bool:
    """Parameters that never receive a gradient (``grad is None``) in the
    reference: the dead frequency-query branch (SpectralTransformer.py:103-106,
    112) and the never-called modules (:231, :233, :248, :250)."""
    if ".attn.q1X1_" in name:
        return True
    return name.startswith(("ups_4.", "ups1.", "ups2.", "output1."))


# ---------------------------------------------------------------------------
# Closed-form deterministic weights / inputs (no torch RNG involved)
# ---------------------------------------------------------------------------
def _hash_uniform(n: int, salt: int) -> np.ndarray:
    """n reproducible values in [-1, 1): 64-bit splitmix-style integer hash of
    (index, salt).  Pure integer arithmetic, identical on every platform."""
    with np.errstate(over="ignore"):
        z = (np.arange(n, dtype=np.uint64) + np.uint64(1)) * np.uint64(0x9E3779B97F4A7C15)
        z = z + np.uint64((salt * 0xD1B54A32D192ED03) & 0xFFFFFFFFFFFFFFFF)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return (z >> np.uint64(40)).astype(np.float64) / float(1 << 23) - 1.0


def _salt(name: str) -> int:
    s = 1469598103934665603
    for ch in name.encode():
        s = ((s ^ ch) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    return s


def closed_form_state(dtype=torch.float32, scale: float = 1.0) -> Params:
    """Deterministic stand-in for the reference's random init: conv weights =
    hash-uniform(-1,1) * fan_in**-0.5, biases = 0.1*uniform, LayerNorm weight =
    1 + 0.1*uniform, LayerNorm bias = 0.1*uniform, temperature = 1 + 0.25*u."""
    out: Params = {}
    for name, shape in param_shapes().items():
        n = int(np.prod(shape))
        u = _hash_uniform(n, _salt(name))
        if name.endswith("temperature"):
            v = 1.0 + 0.25 * u
        elif ".norm" in name and name.endswith("weight"):
            v = 1.0 + 0.1 * u
        elif name.endswith("bias"):
            v = 0.1 * u
        else:
            fan_in = int(np.prod(shape[1:]))
            v = scale * u / math.sqrt(fan_in)
        out[name] = torch.from_numpy(v.reshape(shape)).to(dtype)
    return out


def closed_form_images(b: int, s: int, salt: int, dtype=torch.float32) -> torch.Tensor:
    """Smooth-ish deterministic image batch in [-1, 1], shape [b,3,s,s]: a
    low-frequency pattern plus hash noise, quantised to 8 bits like a decoded
    PNG after Normalize(0.5, 0.5) (src/DataManipulation/DataLoader.py:13-20)."""
    n = b * 3 * s * s
    u = _hash_uniform(n, salt).reshape(b, 3, s, s)
    yy, xx = np.meshgrid(np.arange(s), np.arange(s), indexing="ij")
    base = np.stack([np.sin(2 * np.pi * (yy * (1 + c) + xx * (2 - c) + 7 * salt) / s) for c in range(3)])
    img = 0.5 + 0.3 * base[None] + 0.2 * u
    img = np.clip(np.round(img * 255.0), 0, 255) / 255.0
    return torch.from_numpy((img - 0.5) / 0.5).to(dtype)


def module_input(cin: int, hw: int, salt: int, dtype=torch.float32) -> torch.Tensor:
    """Generic [2, cin, hw, hw] module-test input: pure hash noise in [-1, 1).  (Structured
    inputs - e.g. the sinusoids of closed_form_images - put many spectral bins exactly on the
    real axis, where torch.angle()'s branch cut makes even fp32-vs-fp64 runs of the REFERENCE
    differ by ~1e-3; noise inputs keep module fixtures well-conditioned.)"""
    return torch.from_numpy(_hash_uniform(2 * cin * hw * hw, 5000 + salt).reshape(2, cin, hw, hw)).to(dtype)


# ---------------------------------------------------------------------------
# Live modules
# ---------------------------------------------------------------------------
def layernorm_c(x: torch.Tensor, w: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """LayerNorm over the channel axis of an NCHW tensor, eps 1e-5 (what the
    reshape/transpose dance of SpectralTransformer.py:144-147 computes)."""
    t = F.layer_norm(x.permute(0, 2, 3, 1), (x.shape[1],), w, b, 1e-5)
    return t.permute(0, 3, 1, 2)


def mdta(x: torch.Tensor, P: Params, pfx: str, heads: int) -> torch.Tensor:
    """Live path of MDTA.forward (SpectralTransformer.py:93-100, 107, 110, 113).
    The fftn/gelu/q1X1_1/ifftn/qf/attnf branch (:103-106, 108, 111-112) never
    reaches the output and is not evaluated; neither is the kf half of :107."""
    b, c, hh, ww = x.shape
    qkv = F.conv2d(F.conv2d(x, P[f"{pfx}.qkv.weight"]), P[f"{pfx}.qkv_conv.weight"], padding=1, groups=3 * c)
    q, k, v = (t.reshape(b, heads, c // heads, hh * ww) for t in qkv.split(c, dim=1))
    q = F.normalize(q, dim=-1)          # eps 1e-12 (:98)
    k = F.normalize(k, dim=-1)
    attn = torch.softmax((q @ k.transpose(-2, -1)) * P[f"{pfx}.temperature"], dim=-1)   # (:99)
    out = F.conv2d((attn @ v).reshape(b, c, hh, ww), P[f"{pfx}.project_out.weight"])    # (:100)
    kv = F.conv2d(F.conv2d(out, P[f"{pfx}.kv.weight"]), P[f"{pfx}.kv_conv.weight"], padding=1, groups=2 * c)
    vf = kv[:, c:].reshape(b, heads, c // heads, hh * ww)                               # (:107, :110)
    return F.conv2d((attn @ vf).reshape(b, c, hh, ww), P[f"{pfx}.project_outf.weight"])  # (:113)


def gdfn(x: torch.Tensor, P: Params, pfx: str) -> torch.Tensor:
    """GDFN.forward (SpectralTransformer.py:127-130): exact-erf GELU gate."""
    w_in = P[f"{pfx}.project_in.weight"]
    g = F.conv2d(F.conv2d(x, w_in), P[f"{pfx}.conv.weight"], padding=1, groups=w_in.shape[0])
    g1, g2 = g.chunk(2, dim=1)
    return F.conv2d(F.gelu(g1) * g2, P[f"{pfx}.project_out.weight"])


def transformer_block(x: torch.Tensor, P: Params, pfx: str, heads: int) -> torch.Tensor:
    """TransformerBlock.forward (SpectralTransformer.py:142-148)."""
    x = x + mdta(layernorm_c(x, P[f"{pfx}.norm1.weight"], P[f"{pfx}.norm1.bias"]), P, f"{pfx}.attn", heads)
    x = x + gdfn(layernorm_c(x, P[f"{pfx}.norm2.weight"], P[f"{pfx}.norm2.bias"]), P, f"{pfx}.ffn")
    return x


def block_stack(x, P, pfx, n, heads):
    for j in range(n):
        x = transformer_block(x, P, f"{pfx}.{j}", heads)
    return x


def downsample(x: torch.Tensor, P: Params, pfx: str) -> torch.Tensor:
    """DownSample (SpectralTransformer.py:154-158)."""
    return F.pixel_unshuffle(F.conv2d(x, P[f"{pfx}.body.0.weight"], padding=1), 2)


def _fuse(x, P, pfx):
    """conv1x1+bias, LeakyReLU(0.1), conv1x1+bias (SpectralTransformer.py:165-168)."""
    t = F.leaky_relu(F.conv2d(x, P[f"{pfx}.0.weight"], P[f"{pfx}.0.bias"]), 0.1)
    return F.conv2d(t, P[f"{pfx}.2.weight"], P[f"{pfx}.2.bias"])


def spectral_upsample(x: torch.Tensor, P: Params, pfx: str, use_tile: bool = True) -> torch.Tensor:
    """UpSample.forward (SpectralTransformer.py:174-188).

    ``use_tile=True`` follows the reference literally (tile the fused magnitude
    and phase 2x2, ifft2 on 2H x 2W).  ``use_tile=False`` evaluates the exactly
    equivalent form the HIP path uses: ifft2 on H x W written to the even
    (row, col) positions, zeros elsewhere (SURVEY.md section 2b identity)."""
    fx = torch.fft.fft2(x)
    mag = _fuse(torch.abs(fx), P, f"{pfx}.amp_fuse")
    pha = _fuse(torch.angle(fx), P, f"{pfx}.pha_fuse")
    if use_tile:
        mag, pha = torch.tile(mag, (2, 2)), torch.tile(pha, (2, 2))
        o = torch.abs(torch.fft.ifft2(torch.complex(mag * torch.cos(pha), mag * torch.sin(pha))))
    else:
        small = torch.abs(torch.fft.ifft2(torch.complex(mag * torch.cos(pha), mag * torch.sin(pha))))
        n, c, h, w = small.shape
        o = small.new_zeros(n, c, 2 * h, 2 * w)
        o[:, :, ::2, ::2] = small
    return F.conv2d(o, P[f"{pfx}.post.weight"], P[f"{pfx}.post.bias"])


def shuffle_upsample(x, P, pfx):
    """UpSample1 (SpectralTransformer.py:194-198)."""
    return F.pixel_shuffle(F.conv2d(x, P[f"{pfx}.body.0.weight"], padding=1), 2)


def ups(x, P, pfx, use_tile=True):
    """UpS.forward (SpectralTransformer.py:208-210)."""
    both = torch.cat([spectral_upsample(x, P, f"{pfx}.Fups", use_tile), shuffle_upsample(x, P, f"{pfx}.Sups")], 1)
    return F.conv2d(both, P[f"{pfx}.reduce.weight"])


def forward(P: Params, rgb: torch.Tensor, use_tile: bool = True, taps: dict | None = None) -> torch.Tensor:
    """SpectralTransformer.forward (SpectralTransformer.py:254-270).  ``taps``
    (optional dict) receives named intermediates for stage-wise parity tests."""
    def tap(k, v):
        if taps is not None:
            taps[k] = v
        return v
    f0 = tap("embed", F.conv2d(rgb, P["embed_conv_rgb.weight"], padding=1))
    e1 = tap("enc1", block_stack(f0, P, "encoders.0", 2, 1))
    e2 = tap("enc2", block_stack(downsample(e1, P, "down1"), P, "encoders.1", 3, 2))
    e3 = tap("enc3", block_stack(downsample(e2, P, "down2"), P, "encoders.2", 3, 4))
    e4 = tap("enc4", block_stack(downsample(e3, P, "down3"), P, "encoders.3", 4, 8))
    u1 = tap("ups_1", ups(e4, P, "ups_1", use_tile))
    d3 = tap("dec3", block_stack(F.conv2d(torch.cat([u1, e3], 1), P["reduces1.weight"]), P, "decoders.0", 3, 4))
    u2 = tap("ups_2", ups(d3, P, "ups_2", use_tile))
    d2 = tap("dec2", block_stack(F.conv2d(torch.cat([u2, e2], 1), P["reduces2.weight"]), P, "decoders.1", 3, 2))
    u3 = tap("ups_3", ups(d2, P, "ups_3", use_tile))
    fd = tap("dec1", block_stack(torch.cat([u3, e1], 1), P, "decoders.2", 2, 1))
    fr = tap("refine", block_stack(fd, P, "refinement", 4, 1))
    return F.conv2d(F.conv2d(fr, P["outputl.weight"], padding=1), P["output.weight"], padding=1)


# ---------------------------------------------------------------------------
# Losses and metric
# ---------------------------------------------------------------------------
_Y = (0.299, 0.587, 0.114)


def loss_l1_mean(p, t):
    return (p - t).abs().mean()


def loss_color(p, t):
    """ColorLoss (src/Losses/losses.py:209-212): mean of per-channel spatial
    means of d^2 == global mean of d^2."""
    return ((p - t) ** 2).mean(dim=(2, 3)).mean()


def loss_charbonnier(p, t):
    """CharbonnierLoss, eps=1e-3 (src/Losses/losses.py:185, 190-192)."""
    d = p - t
    return torch.sqrt(d * d + 1e-3 * 1e-3).mean()


def loss_luminance(p, t):
    """LuminanceLoss (src/Losses/luminanceLoss.py:16-19)."""
    y = torch.tensor(_Y, dtype=p.dtype).view(1, 3, 1, 1)
    return F.mse_loss((p * y).sum(1, keepdim=True), (t * y).sum(1, keepdim=True))


def loss_by_name(name: str, p: torch.Tensor, t: torch.Tensor) -> torch.Tensor:
    """LossFunction.getloss for the hot-path names (src/Losses/losses.py:54-78,
    80-81, 128-129).  ``L1withColor`` raises AttributeError in the reference
    (:63 uses ``self.luminanceLoss`` which :49 never builds for that name); the
    evident formula of :59-65 is evaluated instead."""
    bc = t.shape[0] * t.shape[1]
    if name == "L1":
        return loss_l1_mean(p, t) / bc
    if name == "L2":
        return F.mse_loss(p, t) / bc
    if name == "charbonnier":
        return loss_charbonnier(p, t)
    if name == "Luminance":
        return loss_luminance(p, t)
    if name == "LuminanceCharbonnier":                 # src/Losses/losses.py:118-123
        return loss_luminance(p, t) + loss_charbonnier(p, t)
    if name == "L1withColor":
        return (0.5 * loss_color(p, t) + 0.25 * loss_l1_mean(p, t) + 0.25 * loss_luminance(p, t)) / bc
    raise ValueError(f"Unsupported loss: {name}")


def psnr(target, pred):
    """torchPSNR (src/ModelTrainer.py:17-21)."""
    d = torch.clamp(pred, 0, 1) - torch.clamp(target, 0, 1)
    return 20 * torch.log10(1 / (d ** 2).mean().sqrt())


# ---------------------------------------------------------------------------
# Train step (src/ModelTrainer.py:78-90, :199, :55)
# ---------------------------------------------------------------------------
def live_names(P: Params) -> List[str]:
    return [n for n in P if not is_dead_param(n)]


def loss_and_grads(P: Params, raw, ref, loss_name="L1", use_tile=True):
    """forward -> loss -> backward.  Returns (loss, out, {name: grad}) with the
    dead parameters absent (their ``.grad`` stays ``None`` in the reference)."""
    leaves = {n: P[n].detach().clone().requires_grad_(True) for n in live_names(P)}
    full = dict(P)
    full.update(leaves)
    out = forward(full, raw, use_tile)
    loss = loss_by_name(loss_name, out, ref)
    grads = torch.autograd.grad(loss, list(leaves.values()))
    return loss.detach(), out.detach(), dict(zip(leaves.keys(), grads))


class AdamClipState:
    """clip_grad_norm_(params, 1.0) then torch.optim.Adam defaults (lr given,
    betas (0.9, 0.999), eps 1e-8, no weight decay) over the live parameters,
    restated element-wise (torch/optim/adam.py single-tensor path)."""

    def __init__(self, P: Params):
        self.m = {n: torch.zeros_like(P[n]) for n in live_names(P)}
        self.v = {n: torch.zeros_like(P[n]) for n in live_names(P)}
        self.t = 0

    def step(self, P: Params, grads: Params, lr: float, max_norm: float = 1.0, weight_decay: float = 0.0) -> float:
        """weight_decay > 0: torch.optim.AdamW (src/ModelTrainer.py:201) - decoupled decay `p *= 1 - lr * wd` before the
        Adam update (torch/optim/adam.py, decoupled_weight_decay branch)."""
        total = torch.sqrt(sum((g.double() ** 2).sum() for g in grads.values())).to(next(iter(grads.values())).dtype)
        coef = torch.clamp(max_norm / (total + 1e-6), max=1.0)
        self.t += 1
        b1, b2, eps = 0.9, 0.999, 1e-8
        bc1, bc2 = 1 - b1 ** self.t, 1 - b2 ** self.t
        for n, g in grads.items():
            g = g * coef
            self.m[n].mul_(b1).add_(g, alpha=1 - b1)
            self.v[n].mul_(b2).addcmul_(g, g, value=1 - b2)
            denom = (self.v[n].sqrt() / math.sqrt(bc2)).add_(eps)
            P[n] = P[n] * (1 - lr * weight_decay) - (lr / bc1) * self.m[n] / denom
        return float(total)


def multistep_lr(base_lr: float, epoch: int) -> float:
    """MultiStepLR(milestones=[1,100,250], gamma=0.25) (src/ModelTrainer.py:55)."""
    return base_lr * (0.25 ** sum(epoch >= m for m in (1, 100, 250)))
