"""UIQM (reference uqim_utils.py:176-190).
CPU: the oracle (oracle/uiqm_ref.py) against the reference's own outputs (tests/golden/uiqm_ref.npz, recorded by
tests/golden/make_uiqm_vectors.py from the reference imported by path).  GPU: the HIP kernels, through the C ABI, against
both."""
import os

import numpy as np
import pytest
import torch

from oracle import uiqm_ref as U

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "uiqm_ref.npz")


def _cases():
    g = np.load(GOLDEN)
    names = sorted({k.rsplit(".", 1)[0] for k in g.files})
    return g, names


def _close(got, want, rel=1e-9, abs_=1e-12):
    if np.isnan(want):
        return np.isnan(got)
    return abs(got - want) <= rel * abs(want) + abs_


def test_oracle_matches_reference_outputs():
    g, names = _cases()
    assert len(names) == 10
    for n in names:
        img = g[n + ".img"]
        if n + ".raises" in g.files:
            with pytest.raises(ValueError, match="math domain error"):
                U.get_uiqm(img)
            continue
        got = U.get_uiqm(img)
        want = g[n + ".uiqm"]
        for a, b, what in zip(got, want, ("uiqm", "uicm", "uism", "uiconm")):
            assert _close(a, b), (n, what, a, b)
        x = img.astype(np.float32)
        rg = x[:, :, 0].reshape(-1) - x[:, :, 1].reshape(-1)
        assert float(U.trimmed_mean(rg)) == float(g[n + ".mu_a_rg"]), n                 # the float32 sequential sum, bit for bit
        sob = U.sobel_mag(x[:, :, 0])
        assert np.array_equal(sob, g[n + ".sobel_r"], equal_nan=True), n                # scipy's sobel + hypot + scaling, bit for bit
        assert _close(U.eme(sob * x[:, :, 0]), float(g[n + ".eme_r"])), n


def test_oracle_known_answers():
    """Closed forms: a constant image has zero colourfulness spread and zero contrast; the trimmed mean drops
    ceil(0.1 K) + 1 smallest and floor(0.1 K) largest values (the reference's off-by-one, uqim_utils.py:24-26)."""
    x = np.arange(20, dtype=np.float32)
    # K = 20: T_L = 2, T_R = 2 -> sum(sorted[3:18]) / 16
    assert float(U.trimmed_mean(x)) == float(np.float32(sum(range(3, 18)) / 16))
    const = np.full((20, 30, 3), 50, np.uint8)
    assert U.uiconm(const.astype(np.float32)) == 0.0
    assert U.uicm(const.astype(np.float32)) == 0.0


gpu = pytest.mark.gpu


@gpu
@pytest.mark.skipif(not torch.cuda.is_available(), reason="needs a GPU")
def test_hip_uiqm_matches_reference_and_oracle():
    from underwater_image_restoration_amd import metrics
    g, names = _cases()
    for n in names:
        img = g[n + ".img"]
        if n + ".raises" in g.files:
            with pytest.raises(ValueError, match="math domain error"):
                metrics.uiqm(torch.from_numpy(img.astype(np.float32))[None].cuda())
            continue
        want = g[n + ".uiqm"]
        for as_u8 in ((True, False) if img.dtype == np.uint8 else (False,)):
            t = torch.from_numpy(img if as_u8 else img.astype(np.float32))[None].cuda()
            got = metrics.uiqm(t).cpu().numpy()[0]
            for a, b, what in zip(got, want, ("uiqm", "uicm", "uism", "uiconm")):
                # uint8-valued images: every float32 partial sum of the reference is exact -> agreement to double rounding;
                # float-valued images: the reference's float32 sequential sum vs the kernel's double sums -> 1e-5
                tol = 1e-9 if img.dtype == np.uint8 else 2e-5
                assert _close(float(a), float(b), rel=tol, abs_=tol), (n, what, as_u8, a, b)


@gpu
@pytest.mark.skipif(not torch.cuda.is_available(), reason="needs a GPU")
def test_hip_uiqm_batch_and_sizes():
    """A batch of different images of one size, sizes the 10 x 10 window and the workgroup tiles do not divide, 1024 x 1024
    (the loader's size), vs the oracle; argument errors."""
    from underwater_image_restoration_amd import metrics
    rng = np.random.RandomState(3)
    for (n, h, w) in ((5, 40, 70), (3, 101, 67), (2, 256, 256), (1, 1024, 1024), (2, 10, 10)):
        imgs = rng.randint(0, 256, (n, h, w, 3)).astype(np.uint8)
        imgs[0, : h // 2] //= 3
        got = metrics.uiqm(torch.from_numpy(imgs).cuda()).cpu().numpy()
        for i in range(n):
            want = U.get_uiqm(imgs[i])
            # up to ~2^23 / 255 pixels every float32 partial sum of the reference's trimmed mean is exact; beyond (1024 x 1024)
            # ITS left-to-right float32 sum rounds (measured 6e-8 relative on UICM) while the kernel sums in double
            tol = 1e-9 if h * w <= 65536 else 1e-6
            for a, b in zip(got[i], want):
                assert _close(float(a), float(b), rel=tol, abs_=tol), (n, h, w, i, got[i], want)
    with pytest.raises(ValueError):
        metrics.uiqm(torch.zeros(1, 9, 40, 3).cuda())           # smaller than one window
    with pytest.raises(ValueError):
        metrics.uiqm(torch.zeros(1, 40, 40, 4).cuda())
    with pytest.raises(RuntimeError):
        metrics.uiqm(torch.zeros(1, 40, 40, 3))                 # CPU tensor: no fallback
