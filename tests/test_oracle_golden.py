"""Pin the CPU oracle (oracle/cpu_ref.py) to vectors recorded from the
reference itself (tests/golden/make_golden.py).  CPU only."""
import os

import numpy as np
import pytest
import torch

from oracle import cpu_ref as O

torch.set_num_threads(8)


def _g(golden_dir, name):
    return np.load(os.path.join(golden_dir, name), allow_pickle=False)


def rel(a, b):
    a = torch.as_tensor(a).double()
    b = torch.as_tensor(b).double()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


def test_manifest_matches_reference_state_dict(golden_dir):
    m = _g(golden_dir, "st_manifest.npz")
    shapes = O.param_shapes()
    assert list(m["names"]) == list(shapes.keys())
    assert [str(s) for s in shapes.values()] == list(m["shapes"])
    assert int(m["n_params"]) == 2430709 == sum(int(np.prod(s)) for s in shapes.values())
    assert len(shapes) == 443
    assert sum(O.is_dead_param(n) for n in shapes) == 63


def _module_input(cin, hw, salt):
    return O.module_input(cin, hw, salt)


MODULES = [
    ("encoders.0.0", 16, 16, 11, lambda x, P: O.transformer_block(x, P, "encoders.0.0", 1)),
    ("encoders.1.0", 32, 16, 12, lambda x, P: O.transformer_block(x, P, "encoders.1.0", 2)),
    ("refinement.0", 32, 16, 13, lambda x, P: O.transformer_block(x, P, "refinement.0", 1)),
    ("encoders.3.0", 128, 8, 14, lambda x, P: O.transformer_block(x, P, "encoders.3.0", 8)),
    ("encoders.0.0.attn", 16, 16, 15, lambda x, P: O.mdta(x, P, "encoders.0.0.attn", 1)),
    ("encoders.1.0.ffn", 32, 16, 16, lambda x, P: O.gdfn(x, P, "encoders.1.0.ffn")),
    ("down1", 16, 16, 17, lambda x, P: O.downsample(x, P, "down1")),
    ("ups_3.Fups", 32, 16, 18, lambda x, P: O.spectral_upsample(x, P, "ups_3.Fups")),
    ("ups_3.Sups", 32, 16, 19, lambda x, P: O.shuffle_upsample(x, P, "ups_3.Sups")),
    ("ups_3", 32, 16, 20, lambda x, P: O.ups(x, P, "ups_3")),
    ("ups_1", 128, 8, 21, lambda x, P: O.ups(x, P, "ups_1")),
    ("ups_2", 64, 12, 33, lambda x, P: O.ups(x, P, "ups_2")),
]


@pytest.mark.parametrize("tag,cin,hw,salt,fn", MODULES, ids=[m[0] for m in MODULES])
def test_modules_match_reference(golden_dir, tag, cin, hw, salt, fn):
    g = _g(golden_dir, "st_modules.npz")
    P = O.closed_form_state()
    x = _module_input(cin, hw, salt).requires_grad_(True)
    y = fn(x, P)
    cot = torch.from_numpy(O._hash_uniform(y.numel(), 777 + salt).reshape(y.shape)).float()
    (gx,) = torch.autograd.grad((y * cot).sum(), x)
    assert rel(y.detach(), g[f"{tag}.y"]) < 2e-6
    assert rel(gx, g[f"{tag}.gx"]) < 2e-5


def test_spectral_upsample_even_grid_identity():
    """ifft2(tile(Z,(2,2))) on 2Hx2W == ifft2(Z) on the even grid, 0 elsewhere."""
    P = O.closed_form_state()
    x = _module_input(32, 16, 18)
    a = O.spectral_upsample(x, P, "ups_3.Fups", use_tile=True)
    b = O.spectral_upsample(x, P, "ups_3.Fups", use_tile=False)
    assert rel(a, b) < 2e-6


def test_full_model_matches_reference(golden_dir):
    g = _g(golden_dir, "st_full64.npz")
    P = O.closed_form_state()
    raw, ref = O.closed_form_images(2, 64, 101), O.closed_form_images(2, 64, 202)
    loss, out, grads = O.loss_and_grads(P, raw, ref, "L1")
    assert rel(out, g["out"]) < 5e-6
    assert abs(float(loss) - float(g["loss_L1"])) < 1e-7
    assert abs(float(O.loss_charbonnier(out, ref)) - float(g["loss_charbonnier"])) < 1e-6
    assert abs(float(O.loss_color(out, ref)) - float(g["loss_color"])) < 1e-6
    assert abs(float(O.loss_luminance(out, ref)) - float(g["loss_luminance"])) < 1e-6
    # "LuminanceCharbonnier" (src/Losses/losses.py:118-123) = the sum of the two recorded reference values
    assert abs(float(O.loss_by_name("LuminanceCharbonnier", out, ref)) - float(g["loss_luminance"]) - float(g["loss_charbonnier"])) < 2e-6
    assert abs(float(O.psnr(ref, out)) - float(g["psnr"])) < 1e-4
    # the reference leaves exactly these .grad = None
    assert sorted(g["grad_none"]) == sorted(n for n in P if O.is_dead_param(n))
    norms = dict(zip(g["grad_names"], g["grad_norms"]))
    assert set(norms) == set(grads)
    for n, gr in grads.items():
        want = norms[n]
        got = float(gr.double().norm())
        assert abs(got - want) <= 2e-4 * max(want, 1e-6) + 1e-10, (n, got, want)
    for k in g.files:
        if k.startswith("grad/"):
            assert rel(grads[k[5:]], g[k]) < 2e-4, k


def test_charbonnier_known_answer():
    """src/Loss.ipynb:45 - Charbonnier(x, x) = eps = 1e-3."""
    x = O.closed_form_images(1, 16, 5)
    assert abs(float(O.loss_charbonnier(x, x)) - 1e-3) < 1e-9
    for n in ("L1", "L2", "Luminance", "L1withColor"):
        assert float(O.loss_by_name(n, x, x)) == 0.0


def test_train_trajectory_matches_reference(golden_dir):
    """BASELINE config 1: 2 epochs x 2 steps of Adam(3e-4)+clip+MultiStepLR."""
    g = _g(golden_dir, "st_train_cfg1.npz")
    P = O.closed_form_state()
    st = O.AdamClipState(P)
    train = [(O.closed_form_images(1, 64, 1000 + i)[0], O.closed_form_images(1, 64, 2000 + i)[0]) for i in range(4)]
    test = [(O.closed_form_images(1, 64, 3000 + i)[0], O.closed_form_images(1, 64, 4000 + i)[0]) for i in range(2)]
    k = 0
    for epoch in range(2):
        lr = O.multistep_lr(3e-4, epoch)
        for bi in range(2):
            r = torch.stack([train[2 * bi][0], train[2 * bi + 1][0]])
            t = torch.stack([train[2 * bi][1], train[2 * bi + 1][1]])
            loss, _, grads = O.loss_and_grads(P, r, t, "L1")
            nrm = st.step(P, grads, lr)
            assert abs(lr - float(g["lr"][k])) < 1e-12
            assert abs(float(loss) - float(g["step_loss"][k])) < 2e-6, (k, float(loss), g["step_loss"][k])
            assert abs(nrm - float(g["step_norm"][k])) < 2e-4 * float(g["step_norm"][k])
            k += 1
        with torch.no_grad():
            r = torch.stack([test[0][0], test[1][0]])
            t = torch.stack([test[0][1], test[1][1]])
            o = O.forward(P, r)
            assert abs(float(O.loss_by_name("L1", o, t)) - float(g["val_loss"][epoch])) < 2e-6
            assert abs(float(O.psnr(t, o)) - float(g["val_psnr"][epoch])) < 0.05
    fin = dict(zip(g["final_names"], g["final_norms"]))
    for n in P:
        assert abs(float(P[n].double().norm()) - fin[n]) <= 1e-5 * max(fin[n], 1e-6), n


def test_fft_zero_sign_convention(golden_dir):
    """torch.fft.fft2 of real input leaves Im = +0.0 at the self-conjugate bins, so
    angle() there is +pi for negative real parts - the HIP DFT pins the same."""
    g = _g(golden_dir, "fft_zero_sign.npz")
    assert all(bool(g[k]) for k in g.files)


def test_adam_clip_active_matches_torch():
    """clip coefficient < 1 path vs torch.optim.Adam + clip_grad_norm_."""
    torch.manual_seed(1)
    ps = [torch.nn.Parameter(torch.randn(5, 3)), torch.nn.Parameter(torch.randn(7))]
    names = ["embed_conv_rgb.weight", "outputl.weight"]
    P = {n: p.detach().clone() for n, p in zip(names, ps)}
    st = O.AdamClipState(P)
    opt = torch.optim.Adam(ps, lr=1e-2)
    for it in range(3):
        gs = [torch.randn_like(p) * 3 for p in ps]
        for p, g_ in zip(ps, gs):
            p.grad = g_.clone()
        n_ref = torch.nn.utils.clip_grad_norm_(ps, 1.0)
        opt.step()
        n_my = st.step(P, dict(zip(names, gs)), 1e-2)
        assert abs(float(n_ref) - n_my) < 1e-5
        for n, p in zip(names, ps):
            assert rel(P[n], p.detach()) < 1e-6
