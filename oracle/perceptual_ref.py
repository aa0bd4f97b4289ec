"""TEST INFRASTRUCTURE ONLY (never imported by the product).  CPU restatement of VGGPerceptualLoss (src/Losses/losses.py:215-255),
TVLoss (:11-28) and the reference's loss mixes around them (:82-117), with the VGG16 weights passed in as a torchvision-format state
dict (the pretrained file is a download the reference performs; the tests use random weights - the ARITHMETIC is what is checked)."""
import torch
import torch.nn.functional as F

from . import cpu_ref, ffl_ref, msssim_ref

_BLOCKS = ((0, 2), (5, 7), (10, 12, 14), (17, 19, 21))          # conv indices of features[:4] | [4:9] | [9:16] | [16:23]
_MEAN, _STD = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)


def random_vgg16_state(seed: int = 0) -> dict:
    """torchvision vgg16().state_dict() layout for features[:23], He-scaled random values (activations stay alive through 10 layers)."""
    g = torch.Generator().manual_seed(seed)
    cfg = {0: (3, 64), 2: (64, 64), 5: (64, 128), 7: (128, 128), 10: (128, 256), 12: (256, 256), 14: (256, 256), 17: (256, 512),
           19: (512, 512), 21: (512, 512)}
    sd = {}
    for i, (cin, cout) in cfg.items():
        sd[f"features.{i}.weight"] = torch.randn(cout, cin, 3, 3, generator=g) * (2.0 / (9 * cin)) ** 0.5
        sd[f"features.{i}.bias"] = 0.05 * torch.randn(cout, generator=g)
    return sd


def perceptual(inp: torch.Tensor, target: torch.Tensor, sd: dict) -> torch.Tensor:
    dt = inp.dtype
    mean, std = torch.tensor(_MEAN, dtype=dt).view(1, 3, 1, 1), torch.tensor(_STD, dtype=dt).view(1, 3, 1, 1)
    x, y = (inp - mean) / std, (target - mean) / std
    x = F.interpolate(x, mode="bilinear", size=(224, 224), align_corners=False)
    y = F.interpolate(y, mode="bilinear", size=(224, 224), align_corners=False)
    loss = 0.0
    for bi, blk in enumerate(_BLOCKS):
        if bi > 0:
            x, y = F.max_pool2d(x, 2, 2), F.max_pool2d(y, 2, 2)
        for i in blk:
            w, b = sd[f"features.{i}.weight"].to(dt), sd[f"features.{i}.bias"].to(dt)
            x, y = F.relu(F.conv2d(x, w, b, padding=1)), F.relu(F.conv2d(y, w, b, padding=1))
        loss = loss + F.l1_loss(x, y)
    return loss


def tv(x: torch.Tensor) -> torch.Tensor:
    B, C, H, W = x.shape
    h_tv = ((x[:, :, 1:, :] - x[:, :, :H - 1, :]) ** 2).sum()
    w_tv = ((x[:, :, :, 1:] - x[:, :, :, :W - 1]) ** 2).sum()
    return 2 * (h_tv / (C * (H - 1) * W) + w_tv / (C * H * (W - 1))) / B


def gradient_loss(p, t):
    k = torch.tensor([[0., 1, 0], [1, -4, 1], [0, 1, 0]], dtype=p.dtype).expand(3, 1, 3, 3).contiguous()
    return F.l1_loss(F.conv2d(p, k, groups=3), F.conv2d(t, k, groups=3))


def loss(name: str, p: torch.Tensor, t: torch.Tensor, sd: dict):
    """-> the loss; for fflMix the reference's tuple (loss, charb, perc, grad, ffl, ssim)"""
    ch = lambda: cpu_ref.loss_by_name("charbonnier", p, t)
    pe = lambda: perceptual(p, t, sd)
    ss = lambda: 1 - msssim_ref.ms_ssim(p, t)
    if name == "perceptual":
        return pe()
    if name == "mix":
        return 0.03 * ch() + 0.025 * pe() + 0.02 * gradient_loss(p, t) + 0.01 * ss()
    if name == "fflMixTV":
        return 0.03 * ch() + 0.025 * pe() + 0.02 * gradient_loss(p, t) + 0.01 * ss() + 0.02 * ffl_ref.ffl(p, t) + tv(p)
    if name == "bigMix":
        return 0.4 * ch() + 0.25 * pe() + 0.25 * gradient_loss(p, t) + 0.1 * ss()
    if name == "fflMix":
        c, q, g, f, s = ch(), pe(), gradient_loss(p, t), ffl_ref.ffl(p, t), ss()
        return 0.03 * c + 0.025 * q + 0.01 * g + 0.005 * f + 0.1 * s, c, q, g, f, s
    raise ValueError(name)
