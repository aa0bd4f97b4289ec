"""Round 3: the reference losses "gradient" (Gradient_Loss, src/Losses/losses.py:162-181), "AdaptiveLuminanceCharbonnier" (:130-140)
and "LuminanceCharbonnier" (:118-123) as HIP reductions, against values and input gradients recorded from the reference's own
LossFunction (tests/golden/make_loss_vectors.py -> loss_ref.npz).  Tolerance: 2e-6 relative on the loss, 2e-6 of the largest
gradient entry (fp32, other summation order)."""
import os
import numpy as np
import pytest
import torch

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "loss_ref.npz")


def recipe(shape, seed):      # the generator's integer recipe
    n = int(np.prod(shape))
    i = np.arange(n, dtype=np.uint64)
    h = (i * np.uint64(2654435761) + np.uint64(seed) * np.uint64(40503)) % np.uint64(2 ** 32)
    h = (h ^ (h >> np.uint64(15))) * np.uint64(2246822519) % np.uint64(2 ** 32)
    h = (h ^ (h >> np.uint64(13))) % np.uint64(2 ** 32)
    return torch.from_numpy((h.astype(np.float64) / 2 ** 31 - 1.0).astype(np.float32).reshape(shape))


def test_fixture_is_consistent_with_plain_torch():
    """(CPU) the recorded values equal the formulas restated with torch ops - guards the fixture against a bad stand-in import."""
    import torch.nn.functional as F
    z = np.load(GOLD)
    for tag in ("a", "b"):
        shape = tuple(int(v) for v in z[f"{tag}.shape"])
        pred, truth = recipe(shape, 11), recipe(shape, 12)
        truth = 0.7 * truth + 0.3 * pred
        k = torch.tensor([[0., 1, 0], [1, -4, 1], [0, 1, 0]]).expand(3, 1, 3, 3).contiguous()
        g = (F.conv2d(pred, k, groups=3) - F.conv2d(truth, k, groups=3)).abs().mean()
        assert abs(float(g) - float(z[f"{tag}.gradient.loss"])) < 1e-5
        d = pred - truth
        ch = torch.sqrt(d * d + 1e-6).mean()
        y = (d * torch.tensor([0.299, 0.587, 0.114]).view(1, 3, 1, 1)).sum(1)
        lum = (y * y).mean()
        assert abs(float(ch + lum) - float(z[f"{tag}.LuminanceCharbonnier.loss"])) < 1e-6
        w = torch.softmax(-torch.log(torch.stack([ch, lum]) + 1e-6), 0) * 0.5
        assert abs(float((torch.stack([ch, lum]) * w).sum()) - float(z[f"{tag}.AdaptiveLuminanceCharbonnier.loss"])) < 1e-6


@pytest.mark.gpu
@pytest.mark.skipif(not torch.cuda.is_available(), reason="needs a GPU")
@pytest.mark.parametrize("name", ["gradient", "AdaptiveLuminanceCharbonnier", "LuminanceCharbonnier"])
def test_hip_losses_match_the_references_own_values(name):
    from underwater_image_restoration_amd.Losses.losses import LossFunction
    z = np.load(GOLD)
    for tag in ("a", "b"):
        shape = tuple(int(v) for v in z[f"{tag}.shape"])
        pred0, truth = recipe(shape, 11), recipe(shape, 12)
        truth = (0.7 * truth + 0.3 * pred0).cuda()
        pred = pred0.cuda().requires_grad_(True)
        loss = LossFunction(name, "cuda").getloss(pred, truth)
        loss.backward()
        want, wg = float(z[f"{tag}.{name}.loss"]), torch.from_numpy(z[f"{tag}.{name}.grad"])
        assert abs(float(loss) - want) <= 2e-6 * abs(want), (tag, name, float(loss), want)
        err = float((pred.grad.cpu() - wg).abs().max())
        assert err <= 2e-6 * float(wg.abs().max()), (tag, name, err)


@pytest.mark.gpu
@pytest.mark.skipif(not torch.cuda.is_available(), reason="needs a GPU")
def test_unavailable_reference_losses_say_why():
    from underwater_image_restoration_amd.Losses.losses import LossFunction
    x = torch.zeros(1, 3, 16, 16, device="cuda")
    with pytest.raises(NotImplementedError, match="VGG16"):
        LossFunction("fflMix", "cuda").getloss(x, x)
    with pytest.raises(ValueError, match="Unsupported loss"):
        LossFunction("nope", "cuda").getloss(x, x)


# ------------------------------------------------------------------ FocalFrequencyLoss terms (restated package: parity unpinned)
def test_ffl_oracle_properties():
    """(CPU) the restatement behaves like the published loss: zero for equal inputs (0/0 weights -> 0), quadratic in the error's
    scale (the spectrum weights are scale-invariant), invariant to a constant offset of BOTH images, and its weights are detached
    (gradient = 2 w D / count through the orthonormal transform)."""
    from oracle import ffl_ref
    g = torch.Generator().manual_seed(3)
    p, t = torch.rand(2, 3, 12, 20, generator=g, dtype=torch.float64), torch.rand(2, 3, 12, 20, generator=g, dtype=torch.float64)
    assert float(ffl_ref.ffl(p, p)) == 0.0
    l1, l3 = float(ffl_ref.ffl(p, t)), float(ffl_ref.ffl(t + 3.0 * (p - t), t))
    assert abs(l3 - 9.0 * l1) < 1e-12 * l3
    assert abs(float(ffl_ref.ffl(p + 0.25, t + 0.25)) - l1) < 1e-12
    pr = p.clone().requires_grad_(True)
    (gr,) = torch.autograd.grad(ffl_ref.ffl(pr, t), pr)
    d = torch.fft.fft2(p - t, norm="ortho")
    w = d.abs() / d.abs().amax(dim=(-2, -1), keepdim=True)
    want = torch.fft.ifft2(2.0 * w * d / p.numel(), norm="ortho").real
    assert float((gr - want).abs().max()) < 1e-12


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(2, 3, 32, 48), (1, 3, 256, 256), (3, 3, 20, 64), (1, 3, 512, 512)])
@pytest.mark.parametrize("name", ["fflCharbonnier", "LuminanceFFL", "AdaptiveLuminanceFFL"])
def test_ffl_losses_match_the_cpu_restatement(name, shape):
    """HIP value and input gradient of the three FocalFrequencyLoss losses vs oracle/ffl_ref.py in fp64 (the fast MFMA DFT lengths, the
    256 = 2 x 128 split, 512 = 2 x (2 x 128) and the dense fallback)."""
    from oracle import ffl_ref
    from underwater_image_restoration_amd.Losses.losses import LossFunction
    pred, truth = recipe(shape, 21), recipe(shape, 22)
    truth = 0.6 * truth + 0.4 * pred
    p64 = pred.double().requires_grad_(True)
    want = ffl_ref.loss(name, p64, truth.double())
    (gwant,) = torch.autograd.grad(want, p64)
    pg = pred.cuda().requires_grad_(True)
    got = LossFunction(name, "cuda").getloss(pg, truth.cuda())
    got.backward()
    assert abs(float(got) - float(want)) < 5e-6 * abs(float(want)), (float(got), float(want))
    err = float((pg.grad.cpu().double() - gwant).abs().max())
    assert err < 2e-5 * float(gwant.abs().max()), (err, float(gwant.abs().max()))


# ------------------------------------------------------------------ MS-SSIM terms (restated package: parity unpinned)
def test_msssim_oracle_properties():
    """(CPU) the restatement: 1 for identical images, symmetric, below 1 otherwise, and equal to its single-level SSIM factors
    recombined by hand."""
    from oracle import msssim_ref as R
    g = torch.Generator().manual_seed(5)
    x = torch.rand(1, 3, 176, 208, generator=g, dtype=torch.float64)      # sides that halve evenly four times
    y = (x + 0.2 * torch.rand(1, 3, 176, 208, generator=g, dtype=torch.float64)).clamp(0, 1)
    assert abs(float(R.ms_ssim(x, x)) - 1.0) < 1e-12
    a, b = float(R.ms_ssim(x, y)), float(R.ms_ssim(y, x))
    assert abs(a - b) < 1e-12 and 0.0 < a < 1.0
    gw = R._gauss(torch.float64)
    X, Y, f = x, y, []
    for i in range(5):
        s, c = R._ssim(X, Y, gw)
        f.append((s if i == 4 else c).clamp_min(0))
        X, Y = torch.nn.functional.avg_pool2d(X, 2), torch.nn.functional.avg_pool2d(Y, 2)
    want = torch.stack([v ** w for v, w in zip(f, R.WEIGHTS)]).prod(0).mean()
    assert abs(float(want) - a) < 1e-12


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(2, 3, 176, 192), (1, 3, 256, 256), (1, 3, 163, 171)])
@pytest.mark.parametrize("name", ["ms_ssim", "ssimFFLCharbonnier"])
def test_msssim_losses_match_the_cpu_restatement(name, shape):
    """HIP value and input gradient of the two MS-SSIM losses vs oracle/msssim_ref.py in fp64 (even sizes, and odd sizes that take
    the padded average pooling)."""
    from oracle import msssim_ref as R
    from underwater_image_restoration_amd.Losses.losses import LossFunction
    pred = 0.5 + 0.5 * recipe(shape, 31)
    truth = (0.8 * pred + 0.2 * (0.5 + 0.5 * recipe(shape, 32)))
    p64 = pred.double().requires_grad_(True)
    want = R.loss(name, p64, truth.double())
    (gwant,) = torch.autograd.grad(want, p64)
    pg = pred.cuda().requires_grad_(True)
    got = LossFunction(name, "cuda").getloss(pg, truth.cuda())
    got.backward()
    assert abs(float(got) - float(want)) < 2e-5 * max(abs(float(want)), 1e-3), (float(got), float(want))
    err = float((pg.grad.cpu().double() - gwant).abs().max())
    assert err < 1e-4 * float(gwant.abs().max()), (err, float(gwant.abs().max()))


# ------------------------------------------------------------------ VGG perceptual loss and the mixes around it
def test_perceptual_losses_refuse_without_the_users_weights(monkeypatch, tmp_path):
    """(CPU) no weight file, no fake: the names that contain VGGPerceptualLoss say what is missing and where it is looked for."""
    from underwater_image_restoration_amd.Losses import perceptual
    monkeypatch.setattr(perceptual, "_weights_sd", None)
    monkeypatch.setenv("TORCH_HOME", str(tmp_path))
    monkeypatch.delenv("UIR_VGG16_WEIGHTS", raising=False)
    with pytest.raises(NotImplementedError, match="vgg16-397923af.pth"):
        perceptual._find_weights()
    with pytest.raises(KeyError):
        perceptual.set_vgg16_weights({"features.0.weight": torch.zeros(64, 3, 3, 3)})


@pytest.mark.gpu
@pytest.mark.parametrize("name,shape", [("perceptual", (1, 3, 64, 80)), ("perceptual", (2, 3, 256, 240)), ("fflMix", (1, 3, 176, 192)),
                                        ("fflMixTV", (1, 3, 176, 192)), ("bigMix", (1, 3, 176, 192)), ("mix", (1, 3, 176, 192))])
def test_perceptual_mixes_match_the_cpu_restatement(name, shape):
    """Preprocessing, the VGG16 feature stack (random weights in torchvision's layout), the four L1 feature losses, their backward to
    the image, and the reference's mixes (fflMix as its 6-tuple) vs oracle/perceptual_ref.py in fp64.  The L1 feature losses have sign
    kinks: the gradient is compared in the L2 sense."""
    from oracle import perceptual_ref as R
    from underwater_image_restoration_amd.Losses import perceptual
    from underwater_image_restoration_amd.Losses.losses import LossFunction
    sd = R.random_vgg16_state(7)
    perceptual.set_vgg16_weights(sd)
    try:
        pred = 0.5 + 0.45 * recipe(shape, 41)
        truth = 0.7 * pred + 0.3 * (0.5 + 0.45 * recipe(shape, 42))
        p64 = pred.double().requires_grad_(True)
        want = R.loss(name, p64, truth.double(), {k: v.double() for k, v in sd.items()})
        wtot = want[0] if isinstance(want, tuple) else want
        (gwant,) = torch.autograd.grad(wtot, p64)
        pg = pred.cuda().requires_grad_(True)
        got = LossFunction(name, "cuda").getloss(pg, truth.cuda())
        gtot = got[0] if isinstance(got, tuple) else got
        gtot.backward()
        if name == "fflMix":
            assert len(got) == 6
            for a, b in zip(got, want):
                assert abs(float(a) - float(b)) < 2e-4 * max(abs(float(b)), 1e-3), (float(a), float(b))
        assert abs(float(gtot) - float(wtot)) < 2e-4 * abs(float(wtot)), (float(gtot), float(wtot))
        diff = pg.grad.cpu().double() - gwant
        # a fraction f of feature differences within fp32 noise of zero flips its sign between fp32 and fp64: ~2 sqrt(f) in this norm
        # (measured 2e-3 ... 6e-3); a wrong convolution / pooling / resize adjoint is an O(1) error
        assert float(diff.norm()) < 2e-2 * float(gwant.norm()), (float(diff.norm()), float(gwant.norm()))
    finally:
        perceptual._weights_sd = None
        perceptual._VggPlan._cache.clear()
