"""Stand-ins for the third-party names the reference's src/model/*.py import at module level and that are absent from the build
container (build-container-only helper of the make_*.py generators; nothing here is a fixture or reference source):
  timm.layers.{trunc_normal_, to_2tuple, DropPath}   torch.nn.init.trunc_normal_ (the algorithm timm copied from PyTorch), a tuple helper,
                                                     timm's published drop_path function
  torchvision.transforms.functional                  empty (imported at model.py:5, unused by the classes recorded)
  pywt.Wavelet('haar')                               PyWavelets' published Haar filter bank (DWT_2D / IDWT_2D buffers, wave_modules.py:122-161)
"""
import sys
import types

import torch


class DropPath(torch.nn.Module):
    def __init__(self, drop_prob=0., scale_by_keep=True):
        super().__init__()
        self.drop_prob, self.scale_by_keep = drop_prob, scale_by_keep

    def forward(self, x):
        if self.drop_prob == 0. or not self.training:
            return x
        keep = 1 - self.drop_prob
        mask = x.new_empty((x.shape[0],) + (1,) * (x.ndim - 1)).bernoulli_(keep)
        if keep > 0.0 and self.scale_by_keep:
            mask.div_(keep)
        return x * mask


class _Haar:
    dec_lo = [0.7071067811865476, 0.7071067811865476]
    dec_hi = [-0.7071067811865476, 0.7071067811865476]
    rec_lo = [0.7071067811865476, 0.7071067811865476]
    rec_hi = [0.7071067811865476, -0.7071067811865476]

    def __init__(self, name):
        assert name == "haar"


def install():
    timm, layers = types.ModuleType("timm"), types.ModuleType("timm.layers")
    layers.trunc_normal_ = torch.nn.init.trunc_normal_
    layers.to_2tuple = lambda x: x if isinstance(x, tuple) else (x, x)
    layers.DropPath = DropPath
    timm.layers = layers
    tv, tvt, tvf = types.ModuleType("torchvision"), types.ModuleType("torchvision.transforms"), types.ModuleType("torchvision.transforms.functional")
    tv.transforms, tvt.functional = tvt, tvf
    pywt = types.ModuleType("pywt")
    pywt.Wavelet = _Haar
    for name, mod in (("timm", timm), ("timm.layers", layers), ("pywt", pywt), ("torchvision", tv), ("torchvision.transforms", tvt),
                      ("torchvision.transforms.functional", tvf)):
        sys.modules.setdefault(name, mod)
    if "/root/reference" not in sys.path:
        sys.path.insert(0, "/root/reference")
