"""TEST INFRASTRUCTURE ONLY (never imported by the product).  CPU restatement of pytorch_msssim.MS_SSIM as the reference constructs
it (src/Losses/losses.py:43-44: ``MS_SSIM(win_size=11, win_sigma=1.5, data_range=1, size_average=True, channel=3)``).  The package is
an unpinned requirement (requirements.txt:14) that is neither vendored in /root/reference nor installed here: PARITY UNPINNED - this
follows the published implementation (v1.0.0: ``_fspecial_gauss_1d``, ``gaussian_filter`` = separable VALID convolution per channel,
``_ssim`` with K = (0.01, 0.03) returning the per-channel means of ssim_map and cs_map, ``ms_ssim`` with five levels, weights
(0.0448, 0.2856, 0.3001, 0.2363, 0.1333), ``avg_pool2d(kernel 2, padding = size % 2)`` between levels, relu on every factor)."""
import torch
import torch.nn.functional as F

from . import cpu_ref, ffl_ref

WEIGHTS = (0.0448, 0.2856, 0.3001, 0.2363, 0.1333)


def _gauss(dtype):
    coords = torch.arange(11, dtype=torch.float32) - 5
    g = torch.exp(-(coords ** 2) / (2 * 1.5 ** 2))
    return (g / g.sum()).to(dtype)            # the package builds the window in fp32


def _filter(x, g):
    C = x.shape[1]
    x = F.conv2d(x, g.view(1, 1, -1, 1).repeat(C, 1, 1, 1), groups=C)
    return F.conv2d(x, g.view(1, 1, 1, -1).repeat(C, 1, 1, 1), groups=C)


def _ssim(X, Y, g, data_range=1.0, K=(0.01, 0.03)):
    C1, C2 = (K[0] * data_range) ** 2, (K[1] * data_range) ** 2
    mu1, mu2 = _filter(X, g), _filter(Y, g)
    s1 = _filter(X * X, g) - mu1 * mu1
    s2 = _filter(Y * Y, g) - mu2 * mu2
    s12 = _filter(X * Y, g) - mu1 * mu2
    cs_map = (2 * s12 + C2) / (s1 + s2 + C2)
    ssim_map = ((2 * mu1 * mu2 + C1) / (mu1 * mu1 + mu2 * mu2 + C1)) * cs_map
    return ssim_map.flatten(2).mean(-1), cs_map.flatten(2).mean(-1)


def ms_ssim(X: torch.Tensor, Y: torch.Tensor) -> torch.Tensor:
    assert min(X.shape[-2:]) > (11 - 1) * 2 ** 4
    g = _gauss(X.dtype)
    mcs = []
    for i in range(5):
        ssim_pc, cs = _ssim(X, Y, g)
        if i < 4:
            mcs.append(torch.relu(cs))
            pad = [s % 2 for s in X.shape[2:]]
            X, Y = F.avg_pool2d(X, 2, padding=pad), F.avg_pool2d(Y, 2, padding=pad)
    vals = torch.stack(mcs + [torch.relu(ssim_pc)], 0)
    w = torch.tensor(WEIGHTS, dtype=X.dtype).view(-1, 1, 1)
    return torch.prod(vals ** w, dim=0).mean()


def loss(name: str, pred: torch.Tensor, truth: torch.Tensor) -> torch.Tensor:
    if name == "ms_ssim":                       # losses.py:86-87
        return 1 - ms_ssim(pred, truth)
    if name == "ssimFFLCharbonnier":            # :152-156
        return 0.025 * ffl_ref.ffl(pred, truth) + 0.025 * cpu_ref.loss_by_name("charbonnier", pred, truth) + (1 - ms_ssim(pred, truth))
    raise ValueError(name)
