"""CPU oracle for the input stage: PIL bilinear resize -> ToTensor -> Normalize(0.5, 0.5).

TEST INFRASTRUCTURE ONLY (same rule as ``cpu_ref.py``): nothing under
``underwater_image_restoration_amd/`` imports this file.

What the reference does (src/DataManipulation/DataLoader.py:13-20, UIEBDataset.py:27-32): every decoded
RGB image goes through ``transforms.Resize((1024,1024), BILINEAR)`` on a PIL image, i.e. Pillow's
``Image.resize`` (requirements.txt:11 pins ``pillow~=11.1.0``; the container has 12.2.0 -- the resampler
has not changed between them), then ``ToTensor`` (uint8 -> fp32 / 255, CHW) and ``Normalize`` ((x - 0.5) / 0.5).
The arithmetic lives in the third-party package Pillow (``src/libImaging/Resample.c``), which is not vendored in
the reference; this file restates its published algorithm for 8-bit images:

* per axis, ``precompute_coeffs``: triangle filter of support ``max(scale, 1)`` (so down-scaling antialiases),
  window ``[xmin, xmin + n)`` per output index, weights normalised in double precision;
* ``normalize_coeffs_8bpc``: weights rounded half away from zero to 22-bit fixed point;
* horizontal pass over the source rows the vertical pass needs, then the vertical pass; each pass starts its
  int32 accumulator at ``1 << 21``, shifts right by 22 and clamps to [0, 255], so the intermediate image is uint8.

Pin: ``tests/golden/make_resize_vectors.py`` runs Pillow itself in the build container and stores inputs and
outputs under ``tests/golden/resize_pil.npz``; ``tests/test_oracle_golden.py`` checks this file against those
vectors bit for bit (and against the installed Pillow directly when it imports).
"""
from __future__ import annotations

import math
from typing import Tuple

import numpy as np

PRECISION_BITS = 32 - 8 - 2


def precompute_coeffs(in_size: int, out_size: int) -> Tuple[int, np.ndarray, np.ndarray]:
    """(ksize, bounds[out][2] = (first source index, tap count), kk[out][ksize] int32 fixed-point weights)."""
    scale = filterscale = float(in_size) / out_size
    if filterscale < 1.0:
        filterscale = 1.0
    support = 1.0 * filterscale                      # bilinear: support 1
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), np.int32)
    kk = np.zeros((out_size, ksize), np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = 0.0 + (xx + 0.5) * scale
        xmin = int(center - support + 0.5)
        if xmin < 0:
            xmin = 0
        xmax = int(center + support + 0.5)
        if xmax > in_size:
            xmax = in_size
        xmax -= xmin
        w = np.zeros(ksize, np.float64)
        ww = 0.0
        for x in range(xmax):
            a = (x + xmin - center + 0.5) * ss
            if a < 0.0:
                a = -a
            v = 1.0 - a if a < 1.0 else 0.0
            w[x] = v
            ww += v
        for x in range(xmax):
            if ww != 0.0:
                w[x] /= ww
        for x in range(ksize):
            p = w[x] * float(1 << PRECISION_BITS)
            kk[xx, x] = int(-0.5 + p) if w[x] < 0 else int(0.5 + p)
        bounds[xx] = (xmin, xmax)
    return ksize, bounds, kk


def _pass(src: np.ndarray, bounds: np.ndarray, kk: np.ndarray, axis: int) -> np.ndarray:
    """One resampling pass along ``axis`` (0 = rows/vertical, 1 = columns/horizontal) of a uint8 [H][W][3] image."""
    out_size, ksize = kk.shape
    src = np.moveaxis(src, axis, 0).astype(np.int64)            # [n_in][other][3]
    acc = np.full((out_size,) + src.shape[1:], 1 << (PRECISION_BITS - 1), np.int64)
    for t in range(ksize):
        live = t < bounds[:, 1]
        idx = np.where(live, bounds[:, 0] + t, 0)
        wgt = np.where(live, kk[:, t], 0).astype(np.int64)
        acc += src[idx] * wgt[:, None, None]
    out = np.clip(acc >> PRECISION_BITS, 0, 255).astype(np.uint8)
    return np.moveaxis(out, 0, axis)


def resize_bilinear_u8(img: np.ndarray, out_h: int, out_w: int) -> np.ndarray:
    """uint8 [H][W][3] -> uint8 [out_h][out_w][3], equal to ``Image.fromarray(img).resize((out_w, out_h), BILINEAR)``."""
    assert img.dtype == np.uint8 and img.ndim == 3 and img.shape[2] == 3
    in_h, in_w = img.shape[:2]
    if (in_h, in_w) == (out_h, out_w):
        return img.copy()
    _, bh, kh = precompute_coeffs(in_w, out_w)
    _, bv, kv = precompute_coeffs(in_h, out_h)
    if in_w != out_w:
        first = int(bv[0, 0])
        last = int(bv[-1, 0] + bv[-1, 1])
        img = _pass(img[first:last], bh, kh, axis=1)             # only the rows the vertical pass reads
        bv = bv.copy()
        bv[:, 0] -= first
    if in_h != out_h:
        img = _pass(img, bv, kv, axis=0)
    return img


def to_tensor_normalize(img_u8: np.ndarray) -> np.ndarray:
    """uint8 HWC -> float32 CHW in [-1, 1]: ToTensor then Normalize(0.5, 0.5) (DataLoader.py:18-19), in fp32."""
    x = img_u8.astype(np.float32) / np.float32(255.0)
    x = (x - np.float32(0.5)) / np.float32(0.5)
    return np.ascontiguousarray(x.transpose(2, 0, 1))


def load_pair_item(img_u8: np.ndarray, size_hw: Tuple[int, int]) -> np.ndarray:
    return to_tensor_normalize(resize_bilinear_u8(img_u8, size_hw[0], size_hw[1]))
