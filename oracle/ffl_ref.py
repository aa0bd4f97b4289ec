"""TEST INFRASTRUCTURE ONLY (never imported by the product).  CPU restatement of FocalFrequencyLoss as the reference constructs it
(src/Losses/losses.py:45-46: ``FFL(loss_weight=1.0, alpha=1.0)``).  The algorithm lives in the third-party package
``focal_frequency_loss`` (requirements.txt:15, unpinned; v0.3.0 is the release whose defaults the call relies on), which is neither
vendored in /root/reference nor installed in this image: PARITY UNPINNED - this restates the published algorithm with the defaults
patch_factor=1, ave_spectrum=False, log_matrix=False, batch_matrix=False:

    freq(x)   = stack(real, imag) of torch.fft.fft2(x, norm="ortho") per (image, channel)
    tmp       = (freq(pred) - freq(target))^2;  dist = tmp_re + tmp_im
    matrix    = sqrt(dist)^alpha / max over (H, W) of it per (image, channel);  NaN -> 0;  clamp to [0, 1];  detached
    loss      = loss_weight * mean(matrix * dist)

and the three reference loss names built on it (losses.py:97-98, 137-151)."""
import torch

from . import cpu_ref


def ffl(pred: torch.Tensor, truth: torch.Tensor) -> torch.Tensor:
    d = torch.fft.fft2(pred, norm="ortho") - torch.fft.fft2(truth, norm="ortho")
    dist = d.real ** 2 + d.imag ** 2
    m = torch.sqrt(dist)
    w = m / m.amax(dim=(-2, -1), keepdim=True)
    w = torch.where(torch.isnan(w), torch.zeros_like(w), w).clamp(0.0, 1.0).detach()
    return (w * dist).mean()


def _adaptive(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    losses = torch.stack([a, b])
    scales = torch.softmax(-torch.log(losses.detach() + 1e-6), dim=0)
    return torch.sum(losses * torch.tensor([0.5, 0.5], dtype=losses.dtype) * scales)


def loss(name: str, pred: torch.Tensor, truth: torch.Tensor) -> torch.Tensor:
    if name == "fflCharbonnier":
        return ffl(pred, truth) + cpu_ref.loss_by_name("charbonnier", pred, truth)
    if name == "LuminanceFFL":
        return ffl(pred, truth) + cpu_ref.loss_by_name("Luminance", pred, truth)
    if name == "AdaptiveLuminanceFFL":
        return _adaptive(ffl(pred, truth), cpu_ref.loss_by_name("Luminance", pred, truth))
    raise ValueError(name)
