"""CPU oracle of the reference's UIQM (Underwater Image Quality Measure) - TEST INFRASTRUCTURE ONLY.

Restates /root/reference/uqim_utils.py (numpy/scipy, Python pixel loops) with vectorised numpy that follows the reference's
arithmetic type by type (numpy 2.x promotion rules, which the reference pins: requirements.txt `numpy~=2.2.3`):

  getUIQM(x)            uqim_utils.py:176-190   x.astype(float32); 0.0282*UICM + 0.2953*UISM + 3.5753*UIConM
  _uicm                 :36-48   RG = R-G, YB = (R+G)/2-B (float32); asymmetric alpha-trimmed means mu_a (:10-28: sorted, the
                                 sum runs over sorted[ceil(.1K)+1 : K-floor(.1K)] - one element more is dropped on the left than
                                 the formula says - accumulated SEQUENTIALLY IN FLOAT32 by Python's sum() of np.float32 scalars,
                                 weight 1/(K-T_L-T_R) as a Python float times np.float32 -> np.float32); variances s_a (:30-34)
                                 of the float32 differences accumulated in double over ALL pixels
  sobel / eme / _uism   :50-113  scipy.ndimage.sobel(axis 0), (axis 1) in 'reflect' mode on float32 (double line arithmetic,
                                 float32 result per 1-D pass), np.hypot (float32), mag *= 255/max (float32), edge map = mag *
                                 channel, EME over 10x10 blocks: sum log(max/min) (skipped where min == 0 or max == 0), weight
                                 2/(k1*k2); lambda = (0.299, 0.587, 0.144)  [0.144 is the reference's value]
  _uiconm               :141-174 per 10x10x3 block: top = max-min, bot = max+min (float32); (top/bot)*log(top/bot), weight -1/(k1*k2)

Pinned by tests/golden/uiqm_ref.npz: outputs of the reference's own getUIQM (imported by file path in the build container by
tests/golden/make_uiqm_vectors.py) on 10 images (uint8 and float-valued, sizes not divisible by 10, a constant channel -> NaN).
Nothing under underwater_image_restoration_amd/ imports this file.
"""
from __future__ import annotations

import math

import numpy as np

C1, C2, C3 = 0.0282, 0.2953, 3.5753
LAMBDA = (0.299, 0.587, 0.144)
WINDOW = 10


def trimmed_mean(x: np.ndarray, alpha_l: float = 0.1, alpha_r: float = 0.1) -> np.float32:
    """mu_a (uqim_utils.py:10-28) of a flat float32 array, the float32 sequential sum included."""
    xs = np.sort(x.astype(np.float32, copy=False), kind="stable")
    K = xs.size
    t_l, t_r = math.ceil(alpha_l * K), math.floor(alpha_r * K)
    weight = 1 / (K - t_l - t_r)
    s, e = int(t_l + 1), int(K - t_r)
    seg = xs[s:e]
    val = np.cumsum(seg, dtype=np.float32)[-1] if seg.size else np.float32(0)   # sum(): left to right, float32 + float32
    return _weak_mul(weight, val)


def _weak_mul(w: float, v: np.float32) -> np.float32:
    """Python float * np.float32 under NEP 50: the Python scalar is 'weak', the product is computed in float32."""
    return np.float32(w) * np.float32(v)


def variance_about(x: np.ndarray, mu) -> float:
    """s_a (uqim_utils.py:30-34): mean of (pixel - mu)^2 with the difference in float32 and the square / sum in double."""
    d = (x.astype(np.float32, copy=False) - np.float32(mu)).astype(np.float64)
    return float(np.sum(d * d) / x.size)


def uicm(x: np.ndarray) -> float:
    R, G, B = (x[:, :, i].reshape(-1) for i in range(3))
    RG = R - G
    YB = ((R + G) / 2) - B
    mu_rg, mu_yb = trimmed_mean(RG), trimmed_mean(YB)
    s_rg, s_yb = variance_about(RG, mu_rg), variance_about(YB, mu_yb)
    l = math.sqrt(math.pow(mu_rg, 2) + math.pow(mu_yb, 2))
    r = math.sqrt(s_rg + s_yb)
    return (-0.0268 * l) + (0.1586 * r)


def _reflect_pad(a: np.ndarray) -> np.ndarray:
    """One-pixel 'reflect' border of scipy.ndimage (d c b a | a b c d | d c b a): the edge sample is repeated."""
    return np.pad(a, 1, mode="symmetric")


def sobel_pair(ch: np.ndarray):
    """ndimage.sobel(ch, 0), ndimage.sobel(ch, 1) for a float32 plane: derivative [-1,0,1] along the axis, smoothing [1,2,1]
    along the other; every 1-D pass is evaluated in double and stored as float32 (scipy's line buffers)."""
    p = _reflect_pad(ch.astype(np.float32)).astype(np.float64)
    H, W = ch.shape
    # axis 0: derivative along rows (for all padded columns), then smoothing along columns
    d0 = (p[2:, :] - p[:-2, :]).astype(np.float32).astype(np.float64)             # [H][W+2]
    dx = (d0[:, :-2] + 2.0 * d0[:, 1:-1] + d0[:, 2:]).astype(np.float32)
    d1 = (p[:, 2:] - p[:, :-2]).astype(np.float32).astype(np.float64)             # [H+2][W]
    dy = (d1[:-2, :] + 2.0 * d1[1:-1, :] + d1[2:, :]).astype(np.float32)
    return dx, dy


def sobel_mag(ch: np.ndarray) -> np.ndarray:
    dx, dy = sobel_pair(ch)
    mag = np.hypot(dx, dy)                                     # float32
    with np.errstate(divide="ignore", invalid="ignore"):
        mag = mag * (np.float32(255.0) / np.max(mag))           # `mag *= 255.0 / np.max(mag)`: weak Python float -> float32
    return mag


def _blocks(a: np.ndarray, w: int):
    k1, k2 = a.shape[1] // w, a.shape[0] // w
    a = a[:w * k2, :w * k1]
    if a.ndim == 2:
        b = a.reshape(k2, w, k1, w).transpose(0, 2, 1, 3).reshape(k2, k1, -1)
    else:
        b = a.reshape(k2, w, k1, w, a.shape[2]).transpose(0, 2, 1, 3, 4).reshape(k2, k1, -1)
    return b, k1, k2


def _domain_check(ratio: np.ndarray, skip: np.ndarray) -> None:
    """math.log raises ValueError('math domain error') for a ratio <= 0 (signed images): so does the reference, so do we."""
    if np.any(~skip & (ratio <= 0.0)):
        raise ValueError("math domain error")


def eme(a: np.ndarray, w: int = WINDOW) -> float:
    b, k1, k2 = _blocks(a, w)
    mx, mn = b.max(axis=2), b.min(axis=2)                      # NaN propagates, as np.max / np.min do in the reference
    with np.errstate(divide="ignore", invalid="ignore"):
        ratio = (mx / mn).astype(np.float64)                    # float32 division, then math.log of it
        term = np.log(ratio)
    skip = (mn == 0.0) | (mx == 0.0)
    _domain_check(ratio, skip)
    term = np.where(skip, 0.0, term)
    # the reference adds block by block, l (columns) outer, k (rows) inner (uqim_utils.py:77-85)
    return (2.0 / (k1 * k2)) * float(np.sum(term.T.reshape(-1)))


def uism(x: np.ndarray) -> float:
    vals = []
    for i in range(3):
        ch = x[:, :, i]
        vals.append(eme(sobel_mag(ch) * ch))
    return (LAMBDA[0] * vals[0]) + (LAMBDA[1] * vals[1]) + (LAMBDA[2] * vals[2])


def uiconm(x: np.ndarray, w: int = WINDOW) -> float:
    b, k1, k2 = _blocks(x, w)
    mx, mn = b.max(axis=2), b.min(axis=2)
    top, bot = mx - mn, mx + mn                                 # float32
    with np.errstate(divide="ignore", invalid="ignore"):
        ratio = (top / bot).astype(np.float64)
        term = ratio * np.log(ratio)
    skip = np.isnan(top) | np.isnan(bot) | (bot == 0.0) | (top == 0.0)
    _domain_check(ratio, skip)
    term = np.where(skip, 0.0, term)
    return (-1.0 / (k1 * k2)) * float(np.sum(term.T.reshape(-1)))


def get_uiqm(x: np.ndarray):
    """-> (uiqm, uicm, uism, uiconm) like getUIQM (uqim_utils.py:176-190); x: [H][W][3], any real dtype."""
    x = np.asarray(x).astype(np.float32)
    a, b, c = uicm(x), uism(x), uiconm(x, WINDOW)
    return (C1 * a) + (C2 * b) + (C3 * c), a, b, c
