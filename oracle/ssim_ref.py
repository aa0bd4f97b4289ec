"""CPU oracle for the validation SSIM.  TEST INFRASTRUCTURE ONLY (same rule as ``cpu_ref.py``).

The reference's validation loop calls ``torchSSIM(ref, out) = pytorch_msssim.ssim(tar, prd, data_range=1.0,
size_average=True)`` (src/ModelTrainer.py:16,23-24,149,156).  ``pytorch_msssim`` is a third-party package
(requirements.txt:14, unpinned), not vendored in the reference and absent from this image, and the reference holds
no test or golden value for it: **PARITY UNPINNED** -- this file restates the package's published algorithm
(``ssim`` / ``_ssim`` / ``gaussian_filter`` / ``_fspecial_gauss_1d`` of pytorch_msssim 1.0) with the same torch ops:

* window ``g = exp(-(arange(11) - 5)^2 / (2 * 1.5^2))``, ``g /= g.sum()`` in fp32;
* ``gaussian_filter``: depthwise ``F.conv2d`` along H then along W, stride 1, NO padding (a dimension shorter than the
  window would be skipped with a warning; the build refuses it);
* ``mu1, mu2, sigma1_sq = G(X*X) - mu1^2, sigma2_sq, sigma12 = G(X*Y) - mu1*mu2``; ``C1 = (0.01 L)^2``, ``C2 = (0.03 L)^2``;
  ``cs_map = (2 sigma12 + C2) / (sigma1_sq + sigma2_sq + C2)``;
  ``ssim_map = (2 mu1 mu2 + C1) / (mu1^2 + mu2^2 + C1) * cs_map``;
* per-channel mean over the valid pixels, then the mean over [N, C] (``size_average=True``); no ReLU (``nonnegative_ssim=False``).
"""
from __future__ import annotations

import torch
import torch.nn.functional as F


def gauss_window(size: int = 11, sigma: float = 1.5) -> torch.Tensor:
    coords = torch.arange(size, dtype=torch.float32)
    coords -= size // 2
    g = torch.exp(-(coords ** 2) / (2 * sigma ** 2))
    g /= g.sum()
    return g


def _gaussian_filter(x: torch.Tensor, win: torch.Tensor) -> torch.Tensor:
    C = x.shape[1]
    w = win.to(x.dtype).reshape(1, 1, 1, -1).repeat(C, 1, 1, 1)
    out = F.conv2d(x, w.transpose(2, -1), stride=1, padding=0, groups=C)      # along H
    return F.conv2d(out, w, stride=1, padding=0, groups=C)                     # along W


def ssim(X: torch.Tensor, Y: torch.Tensor, data_range: float = 1.0) -> torch.Tensor:
    """0-dim tensor = pytorch_msssim.ssim(X, Y, data_range=data_range, size_average=True) for [N,C,H,W] inputs."""
    assert X.shape == Y.shape and X.dim() == 4 and X.shape[2] >= 11 and X.shape[3] >= 11
    win = gauss_window()
    C1, C2 = (0.01 * data_range) ** 2, (0.03 * data_range) ** 2
    mu1, mu2 = _gaussian_filter(X, win), _gaussian_filter(Y, win)
    mu1_sq, mu2_sq, mu1_mu2 = mu1.pow(2), mu2.pow(2), mu1 * mu2
    sigma1_sq = _gaussian_filter(X * X, win) - mu1_sq
    sigma2_sq = _gaussian_filter(Y * Y, win) - mu2_sq
    sigma12 = _gaussian_filter(X * Y, win) - mu1_mu2
    cs_map = (2 * sigma12 + C2) / (sigma1_sq + sigma2_sq + C2)
    ssim_map = ((2 * mu1_mu2 + C1) / (mu1_sq + mu2_sq + C1)) * cs_map
    return torch.flatten(ssim_map, 2).mean(-1).mean()
