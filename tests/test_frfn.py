"""FRFN feed-forward (reference src/model/block.py:249-282).
CPU: the oracle (oracle/frfn_ref.py) against outputs and gradients recorded from the reference's own class
(tests/golden/frfn_ref.npz <- tests/golden/make_frfn_vectors.py).  GPU: the HIP launch plan behind the reference's surface
against both, through autograd like the reference's trainer would drive it."""
import os

import numpy as np
import pytest
import torch

from oracle import frfn_ref as R

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "frfn_ref.npz")
CASES = {"d32_b2_s16": (32, 2, 16), "d64_b1_s8": (64, 1, 8), "d32_b1_s24": (32, 1, 24)}


def _rel(a, b):
    a, b = torch.as_tensor(a).double().cpu(), torch.as_tensor(b).double().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


@pytest.mark.parametrize("case", list(CASES))
def test_oracle_matches_reference_frfn(case):
    g = np.load(GOLDEN)
    dim, B, S = CASES[case]
    assert list(g["names"]) == list(R.param_shapes(32, 128).keys())          # the reference's registration order
    P = {k: v.double() for k, v in R.closed_form_params(dim, 4 * dim).items()}
    x = R.closed_form_input(B, S * S, dim, 17).double()
    probe = R.closed_form_input(B, S * S, dim, 23).double()
    y, dx, grads = R.out_and_grads(x, P, probe)
    assert _rel(y, g[case + ".y"]) < 1e-6 and _rel(dx, g[case + ".dx"]) < 1e-6          # fixtures are float32 of the fp64 run
    for n, gw in grads.items():
        assert _rel(gw, g[case + ".grad." + n]) < 1e-6, n


gpu = pytest.mark.gpu


@gpu
@pytest.mark.skipif(not torch.cuda.is_available(), reason="needs a GPU")
@pytest.mark.parametrize("case", list(CASES))
def test_hip_frfn_matches_reference(case):
    """Output, input gradient and all seven parameter gradients vs the reference's own (north-star tolerance 1e-3 relative;
    the reference's fp32 run is itself 1e-7 from its fp64 run on these inputs)."""
    from underwater_image_restoration_amd.model.block import FRFN
    g = np.load(GOLDEN)
    dim, B, S = CASES[case]
    m = FRFN(dim, 4 * dim)
    m.load_state_dict(R.closed_form_params(dim, 4 * dim))
    m = m.cuda()
    x = R.closed_form_input(B, S * S, dim, 17).cuda().requires_grad_(True)
    probe = R.closed_form_input(B, S * S, dim, 23).cuda()
    y = m(x)
    (y * probe).sum().backward()
    torch.cuda.synchronize()
    e_y, e_dx = _rel(y, g[case + ".y"]), _rel(x.grad, g[case + ".dx"])
    worst = max(_rel(p.grad, g[case + ".grad." + n]) for n, p in m.named_parameters())
    print(f"{case}: y {e_y:.2e} dx {e_dx:.2e} worst param grad {worst:.2e}")
    assert e_y < 2e-5 and e_dx < 2e-5 and worst < 1e-4
    with torch.no_grad():
        assert torch.equal(m(x.detach()), y.detach())                      # the forward-only plan computes the same values


@gpu
@pytest.mark.skipif(not torch.cuda.is_available(), reason="needs a GPU")
def test_hip_frfn_full_size_properties_and_surface():
    """cfg4's first level (dim 32, hidden 128) at 256 x 256, B = 2 vs the fp64 oracle on a 2-sample batch (seconds on the
    CPU); run-to-run bitwise determinism; state_dict names / shapes of the reference; argument errors."""
    from underwater_image_restoration_amd.model.block import FRFN
    dim, B, S = 32, 2, 256
    m = FRFN(dim, 4 * dim)
    assert {k: tuple(v.shape) for k, v in m.state_dict().items()} == R.param_shapes(dim, 4 * dim)
    P = R.closed_form_params(dim, 4 * dim)
    m.load_state_dict(P)
    m = m.cuda()
    x = R.closed_form_input(B, S * S, dim, 31)
    probe = R.closed_form_input(B, S * S, dim, 32)
    y64, dx64, g64 = R.out_and_grads(x.double(), {k: v.double() for k, v in P.items()}, probe.double())
    outs = []
    for _ in range(2):
        xg = x.cuda().requires_grad_(True)
        for p in m.parameters():
            p.grad = None
        y = m(xg)
        (y * probe.cuda()).sum().backward()
        torch.cuda.synchronize()
        outs.append((y.detach().clone(), xg.grad.clone(), [p.grad.clone() for p in m.parameters()]))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    assert all(torch.equal(a, b) for a, b in zip(outs[0][2], outs[1][2]))
    assert _rel(outs[0][0], y64) < 2e-5 and _rel(outs[0][1], dx64) < 2e-5
    for (n, _), gw in zip(m.named_parameters(), outs[0][2]):
        assert _rel(gw, g64[n]) < 2e-4, n
    with pytest.raises(ValueError):
        m(torch.zeros(1, 15, dim).cuda())                                  # not a square token grid
    with pytest.raises(RuntimeError):
        FRFN(32, 128)(torch.zeros(1, 16, 32))                              # CPU tensor: no fallback


# ---------------------------------------------------------------------------------------------- LeFF (block.py:223-247)
LEFF_CASES = {"leff_d32_b2_s16": (32, 2, 16), "leff_d64_b1_s8": (64, 1, 8)}


@pytest.mark.parametrize("case", list(LEFF_CASES))
def test_oracle_matches_reference_leff(case):
    g = np.load(GOLDEN)
    dim, B, S = LEFF_CASES[case]
    assert list(g["leff_names"]) == list(R.leff_param_shapes(32, 128).keys())
    P = {k: v.double() for k, v in R.closed_form_leff_params(dim, 4 * dim).items()}
    x, probe = R.closed_form_input(B, S * S, dim, 19).double(), R.closed_form_input(B, S * S, dim, 29).double()
    y, dx, grads = R.leff_out_and_grads(x, P, probe)
    assert _rel(y, g[case + ".y"]) < 1e-6 and _rel(dx, g[case + ".dx"]) < 1e-6
    for n, gw in grads.items():
        assert _rel(gw, g[case + ".grad." + n]) < 1e-6, n


@gpu
@pytest.mark.skipif(not torch.cuda.is_available(), reason="needs a GPU")
@pytest.mark.parametrize("case", list(LEFF_CASES))
def test_hip_leff_matches_reference(case):
    from underwater_image_restoration_amd.model.block import LeFF
    g = np.load(GOLDEN)
    dim, B, S = LEFF_CASES[case]
    m = LeFF(dim, 4 * dim)
    assert list(m.state_dict().keys()) == list(g["leff_names"])
    m.load_state_dict(R.closed_form_leff_params(dim, 4 * dim))
    m = m.cuda()
    x = R.closed_form_input(B, S * S, dim, 19).cuda().requires_grad_(True)
    probe = R.closed_form_input(B, S * S, dim, 29).cuda()
    y = m(x)
    (y * probe).sum().backward()
    torch.cuda.synchronize()
    worst = max(_rel(p.grad, g[case + ".grad." + n]) for n, p in m.named_parameters())
    print(f"{case}: y {_rel(y, g[case + '.y']):.2e} dx {_rel(x.grad, g[case + '.dx']):.2e} worst param grad {worst:.2e}")
    assert _rel(y, g[case + ".y"]) < 2e-5 and _rel(x.grad, g[case + ".dx"]) < 2e-5 and worst < 1e-4


# ---------------------------------------------------------------------------------------------- FDFP (block.py:518-556)
FDFP_CASES = {"fdfp_d32_b2_s16": (32, 2, 16), "fdfp_d64_b1_s8": (64, 1, 8)}


@pytest.mark.parametrize("case", list(FDFP_CASES))
def test_oracle_matches_reference_fdfp(case):
    g = np.load(GOLDEN)
    dim, B, S = FDFP_CASES[case]
    P = {k: v.double() for k, v in R.closed_form_fdfp_params(dim, 2 * dim).items()}
    x = R.closed_form_input(B, S * S, dim, 37).double().reshape(B, S, S, dim)
    probe = R.closed_form_input(B, S * S, dim, 39).double().reshape(B, S, S, dim)
    y, dx, grads = R.fdfp_out_and_grads(x, P, probe)
    assert _rel(y, g[case + ".y"]) < 1e-6 and _rel(dx, g[case + ".dx"]) < 1e-6
    for n, gw in grads.items():
        assert _rel(gw, g[case + ".grad." + n]) < 1e-6, n


def test_fdfp_state_dict_is_the_references():
    """Keys in the reference's order, including the Haar buffers its constructor registers even in Fourier mode, bit for bit."""
    from underwater_image_restoration_amd.model.block import FDFP
    g = np.load(GOLDEN)
    sd = FDFP(32, 64).state_dict()
    assert list(sd.keys()) == list(g["fdfp_state_keys"])
    for k, v in sd.items():
        if k.startswith(("dwt.", "idwt.")):
            assert np.array_equal(v.numpy(), g["fdfp_buffer." + k]), k


@gpu
@pytest.mark.skipif(not torch.cuda.is_available(), reason="needs a GPU")
@pytest.mark.parametrize("case", list(FDFP_CASES))
def test_hip_fdfp_matches_reference(case):
    from underwater_image_restoration_amd.model.block import FDFP
    g = np.load(GOLDEN)
    dim, B, S = FDFP_CASES[case]
    m = FDFP(dim, 2 * dim)
    m.load_state_dict(R.closed_form_fdfp_params(dim, 2 * dim), strict=False)
    m = m.cuda()
    x = R.closed_form_input(B, S * S, dim, 37).reshape(B, S, S, dim).cuda().requires_grad_(True)
    probe = R.closed_form_input(B, S * S, dim, 39).reshape(B, S, S, dim).cuda()
    y = m(x)
    assert y.shape == x.shape
    (y * probe).sum().backward()
    torch.cuda.synchronize()
    worst = max(_rel(p.grad, g[case + ".grad." + n]) for n, p in m.named_parameters())
    print(f"{case}: y {_rel(y, g[case + '.y']):.2e} dx {_rel(x.grad, g[case + '.dx']):.2e} worst param grad {worst:.2e}")
    assert _rel(y, g[case + ".y"]) < 2e-5 and _rel(x.grad, g[case + ".dx"]) < 5e-5 and worst < 2e-4
    # 64 x 64 (one launch of the two-stage MFMA DFT per axis) and 256 x 256 (each axis factored 2 x 128) vs the fp64 oracle
    for (nb, sz) in (((2, 64), (1, 256)) if dim == 32 else ()):
        P = R.closed_form_fdfp_params(dim, 2 * dim)
        xb = R.closed_form_input(nb, sz * sz, dim, 47).reshape(nb, sz, sz, dim)
        pb = R.closed_form_input(nb, sz * sz, dim, 49).reshape(nb, sz, sz, dim)
        y64, dx64, g64 = R.fdfp_out_and_grads(xb.double(), {k: v.double() for k, v in P.items()}, pb.double())
        xg = xb.cuda().requires_grad_(True)
        for p in m.parameters():
            p.grad = None
        yb = m(xg)
        (yb * pb.cuda()).sum().backward()
        assert _rel(yb, y64) < 5e-5 and _rel(xg.grad, dx64) < 1e-4
        assert max(_rel(p.grad, g64[n]) for n, p in m.named_parameters()) < 5e-4
