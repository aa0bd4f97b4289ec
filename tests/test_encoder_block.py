"""EncoderBlock of the NewBigFRFNModel family in its shipped Fourier mode (reference src/model/model.py:16-93).
CPU: the oracle (oracle/encoder_ref.py) against the reference's own class (tests/golden/encoder_ref.npz <-
tests/golden/make_encoder_vectors.py).  GPU: the HIP launch plan behind the reference's constructor / forward surface."""
import os

import numpy as np
import pytest
import torch

from oracle import encoder_ref as E, frfn_ref as R

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "encoder_ref.npz")
CASES = {"d32_b2_s16": (32, 2, 16), "d64_b1_s8": (64, 1, 8)}


def _rel(a, b):
    a, b = torch.as_tensor(a).detach().double().cpu(), torch.as_tensor(b).detach().double().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


@pytest.mark.parametrize("case", list(CASES))
def test_oracle_matches_reference_encoder_block(case):
    g = np.load(GOLDEN)
    dim, B, S = CASES[case]
    assert list(g["names"]) == list(E.param_shapes(32, 128).keys())
    P = {k: v.double() for k, v in E.closed_form_params(dim, 4 * dim).items()}
    x = R.closed_form_input(B, S * S, dim, 41).double()
    probe = R.closed_form_input(B, S * S, dim, 43).double()
    y, dx, grads = E.out_and_grads(x, P, probe)
    assert _rel(y, g[case + ".y"]) < 1e-6 and _rel(dx, g[case + ".dx"]) < 1e-6
    assert sorted(g[case + ".dead"]) == ["norm2.bias", "norm2.weight"] and all(not n.startswith("norm2") for n in grads)
    for n, gw in grads.items():
        assert _rel(gw, g[case + ".grad." + n]) < 1e-6, n


def test_token_fft_operator_identity():
    """Re fftn over (L, C) of a real tensor = C_L x C_C - S_L x S_C, Re ifftn = the same / (L*C), and the operator is symmetric
    (what the HIP plan relies on for the backward pass)."""
    torch.manual_seed(0)
    Ltok, C = 36, 8
    x, y = torch.randn(Ltok, C, dtype=torch.float64), torch.randn(Ltok, C, dtype=torch.float64)

    def cs(n):
        a = 2 * np.pi * torch.outer(torch.arange(n, dtype=torch.float64), torch.arange(n, dtype=torch.float64)) / n
        return torch.cos(a), torch.sin(a)
    (cl, sl), (cc, sc) = cs(Ltok), cs(C)
    T = lambda v: cl @ v @ cc - sl @ v @ sc                                   # noqa: E731
    assert torch.allclose(T(x), torch.fft.fftn(x, dim=(-2, -1)).real, atol=1e-9)
    assert torch.allclose(T(x) / (Ltok * C), torch.fft.ifftn(x, dim=(-2, -1)).real, atol=1e-12)
    assert abs(float((T(x) * y).sum() - (x * T(y)).sum())) < 1e-8


gpu = pytest.mark.gpu


@gpu
@pytest.mark.skipif(not torch.cuda.is_available(), reason="needs a GPU")
@pytest.mark.parametrize("case", list(CASES))
def test_hip_encoder_block_matches_reference(case):
    from underwater_image_restoration_amd.model.model import EncoderBlock
    g = np.load(GOLDEN)
    dim, B, S = CASES[case]
    m = EncoderBlock(dim=dim, input_resolution=(S, S), num_heads=4, mlp_ratio=4, token_mlp="frfn", drop_path=0.0,
                     freq_mlp="frfn", use_dwt="Fourier")
    assert list(m.state_dict().keys()) == list(g["names"])
    m.load_state_dict(E.closed_form_params(dim, 4 * dim))
    m = m.cuda()
    x = R.closed_form_input(B, S * S, dim, 41).cuda().requires_grad_(True)
    probe = R.closed_form_input(B, S * S, dim, 43).cuda()
    y = m(x)
    (y * probe).sum().backward()
    torch.cuda.synchronize()
    e_y, e_dx = _rel(y, g[case + ".y"]), _rel(x.grad, g[case + ".dx"])
    worst = ("", 0.0)
    for n, p in m.named_parameters():
        if n.startswith("norm2."):
            assert p.grad is None, n                                                  # like the reference: never used
            continue
        e = _rel(p.grad, g[case + ".grad." + n])
        worst = max(worst, (n, e), key=lambda t: t[1])
    print(f"{case}: y {e_y:.2e} dx {e_dx:.2e} worst param grad {worst}")
    assert e_y < 2e-5 and e_dx < 5e-5 and worst[1] < 2e-4


@gpu
@pytest.mark.skipif(not torch.cuda.is_available(), reason="needs a GPU")
def test_hip_encoder_block_full_size_and_surface():
    """cfg4's first level (dim 32) at 256 x 256 (L = 65 536: the two-stage 256 x 256 token transform), B = 2, vs the fp64 oracle;
    bitwise determinism; the reference's constructor errors."""
    from underwater_image_restoration_amd.model.model import EncoderBlock, token_factors
    assert token_factors(256, 256) == (64, 32, 32) and token_factors(64, 64) == (128, 32) and token_factors(16, 16) == (16, 16)
    dim, B = 32, 2
    m = EncoderBlock(dim, (64, 64), 4, token_mlp="frfn", freq_mlp="frfn")
    P = E.closed_form_params(dim, 4 * dim)
    m.load_state_dict(P)
    m = m.cuda()
    # 64 x 64 tokens: the two-stage (128, 32) transform on the MFMA DFT kernel
    xs, ps = R.closed_form_input(B, 64 * 64, dim, 53), R.closed_form_input(B, 64 * 64, dim, 54)
    ys64, dxs64, gs64 = E.out_and_grads(xs.double(), {k: v.double() for k, v in P.items()}, ps.double())
    xg = xs.cuda().requires_grad_(True)
    ys = m(xg)
    (ys * ps.cuda()).sum().backward()
    torch.cuda.synchronize()
    assert _rel(ys, ys64) < 5e-5 and _rel(xg.grad, dxs64) < 1e-4
    assert max(_rel(p.grad, gs64[n]) for n, p in m.named_parameters() if p.grad is not None) < 5e-4
    S = 256                                                  # 65 536 tokens: three stages (64, 32, 32)
    x = R.closed_form_input(B, S * S, dim, 51)
    probe = R.closed_form_input(B, S * S, dim, 52)
    torch.set_num_threads(8)
    y64, dx64, g64 = E.out_and_grads(x.double(), {k: v.double() for k, v in P.items()}, probe.double())
    outs = []
    for _ in range(2):
        xg = x.cuda().requires_grad_(True)
        for p in m.parameters():
            p.grad = None
        y = m(xg)
        (y * probe.cuda()).sum().backward()
        torch.cuda.synchronize()
        outs.append((y.detach().clone(), xg.grad.clone(), {n: p.grad.clone() for n, p in m.named_parameters() if p.grad is not None}))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    assert all(torch.equal(outs[0][2][n], outs[1][2][n]) for n in outs[0][2])
    e_y, e_dx = _rel(outs[0][0], y64), _rel(outs[0][1], dx64)
    worst = max(((n, _rel(gw, g64[n])) for n, gw in outs[0][2].items()), key=lambda t: t[1])
    print(f"256x256: y {e_y:.2e} dx {e_dx:.2e} worst param grad {worst}")
    assert e_y < 1e-4 and e_dx < 2e-4 and worst[1] < 1e-3            # north-star: 1e-3 relative fp32; a length-65536 fp32 transform
    with pytest.raises(ValueError, match="Unknown token_mlp"):
        EncoderBlock(32, (16, 16), 4, token_mlp="nope")
    with pytest.raises(NotImplementedError):
        EncoderBlock(32, (16, 16), 4, token_mlp="frfn", freq_mlp="frfn", use_dwt="Wavelet")


@gpu
@pytest.mark.skipif(not torch.cuda.is_available(), reason="needs a GPU")
def test_hip_encoder_block_leff_vs_oracle():
    """The class default (token_mlp = freq_mlp = "leff", model.py:18-19) and a mixed block, against the fp64 oracle
    (LeFF itself is pinned to the reference's class in tests/test_frfn.py)."""
    import torch.nn.functional as F
    from underwater_image_restoration_amd.model.model import EncoderBlock
    dim, B, S = 32, 2, 16
    for tok, frq in (("leff", "leff"), ("frfn", "leff")):
        m = EncoderBlock(dim, (S, S), 4, token_mlp=tok, freq_mlp=frq)
        P = {}
        for n, p in m.named_parameters():
            is_leff = (n.startswith("mlp.") and tok == "leff") or (n.startswith("freq_mlp.") and frq == "leff")
            base = R.closed_form_leff_params(dim, 4 * dim) if is_leff else R.closed_form_params(dim, 4 * dim)
            key = n.split(".", 1)[1] if n.startswith(("mlp.", "freq_mlp.")) else None
            P[n] = base[key].clone() if key else E.closed_form_params(dim, 4 * dim)[n].clone()
        m.load_state_dict(P)
        m = m.cuda()
        x, probe = R.closed_form_input(B, S * S, dim, 61), R.closed_form_input(B, S * S, dim, 62)

        def sub(pfx):
            return {k[len(pfx):]: v for k, v in leaves.items() if k.startswith(pfx)}
        leaves = {n: v.double().requires_grad_(True) for n, v in P.items()}
        xd = x.double().requires_grad_(True)
        f_tok = R.leff if tok == "leff" else R.frfn
        f_frq = R.leff if frq == "leff" else R.frfn
        x1 = f_tok(F.layer_norm(xd, (dim,), leaves["norm1.weight"], leaves["norm1.bias"], 1e-5), sub("mlp."))
        f = f_frq(torch.fft.fftn(x1, dim=(-2, -1)).real, sub("freq_mlp."))
        y64 = xd + torch.fft.ifftn(f, dim=(-2, -1)).real + x1
        (y64 * probe.double()).sum().backward()
        xg = x.cuda().requires_grad_(True)
        y = m(xg)
        (y * probe.cuda()).sum().backward()
        torch.cuda.synchronize()
        assert _rel(y, y64) < 2e-5 and _rel(xg.grad, xd.grad) < 5e-5, (tok, frq)
        for n, p in m.named_parameters():
            if not n.startswith("norm2."):
                assert _rel(p.grad, leaves[n].grad) < 2e-4, (tok, frq, n)


@gpu
@pytest.mark.skipif(not torch.cuda.is_available(), reason="needs a GPU")
def test_hip_encoder_block_dim_256_vs_oracle():
    """encoder_3 of MyBigFRFNModel (model.py:513-520): dim 256 (LayerNorm over 256, FRFN 256 -> 2048 -> 256), 32 x 32 tokens
    (the (32, 32) two-stage token transform), drop_path 0 here; vs the fp64 oracle."""
    from underwater_image_restoration_amd.model.model import EncoderBlock
    dim, B, S2 = 256, 1, 32
    m = EncoderBlock(dim, (S2, S2), 4, token_mlp="frfn", freq_mlp="frfn")
    P = E.closed_form_params(dim, 4 * dim)
    m.load_state_dict(P)
    m = m.cuda()
    x, probe = R.closed_form_input(B, S2 * S2, dim, 111), R.closed_form_input(B, S2 * S2, dim, 112)
    torch.set_num_threads(8)
    y64, dx64, g64 = E.out_and_grads(x.double(), {k: v.double() for k, v in P.items()}, probe.double())
    xg = x.cuda().requires_grad_(True)
    y = m(xg)
    (y * probe.cuda()).sum().backward()
    torch.cuda.synchronize()
    worst = max(((n, _rel(p.grad, g64[n])) for n, p in m.named_parameters() if p.grad is not None), key=lambda t: t[1])
    print(f"EncoderBlock(256): y {_rel(y, y64):.2e} dx {_rel(xg.grad, dx64):.2e} worst {worst}")
    assert _rel(y, y64) < 1e-4 and _rel(xg.grad, dx64) < 2e-4 and worst[1] < 1e-3


@gpu
@pytest.mark.skipif(not torch.cuda.is_available(), reason="needs a GPU")
def test_hip_encoder_block_drop_path():
    """Stochastic depth (timm's drop_path, model.py:27,54,91): in training mode each branch of every sample is kept with probability
    1 - p and scaled by 1 / (1 - p); eval mode is the identity.  The factors the plan used are read back and fed to the oracle:
    output, input gradient and parameter gradients must match; the draws follow torch's generator (same seed -> same factors)."""
    import torch.nn.functional as F
    from underwater_image_restoration_amd.model.model import EncoderBlock
    dim, B, S2, p = 32, 8, 16, 0.5
    m = EncoderBlock(dim, (S2, S2), 4, token_mlp="frfn", freq_mlp="frfn", drop_path=p)
    P = E.closed_form_params(dim, 4 * dim)
    m.load_state_dict(P)
    m = m.cuda()
    x, probe = R.closed_form_input(B, S2 * S2, dim, 121), R.closed_form_input(B, S2 * S2, dim, 122)
    m.eval()
    with torch.no_grad():
        y_eval = m(x.cuda())
    y_ref, _, _ = E.out_and_grads(x.double(), {k: v.double() for k, v in P.items()}, probe.double())
    assert _rel(y_eval, y_ref) < 2e-5                                  # eval: DropPath is the identity
    m.train()
    torch.manual_seed(7)
    xg = x.cuda().requires_grad_(True)
    y = m(xg)
    (y * probe.cuda()).sum().backward()
    torch.cuda.synchronize()
    plan = m._plan_for(xg)
    s = plan.dp_scale.cpu().double()                                   # [2][B]: row 0 = drop_path2 (frequency), row 1 = drop_path
    assert set(s.unique().tolist()) <= {0.0, 1.0 / (1 - p)} and 0 < int((s == 0).sum()) < 2 * B
    torch.manual_seed(7)                                               # the reference's draw order: drop_path2, then drop_path
    want = torch.stack([torch.empty(B, device="cuda").bernoulli_(1 - p).div_(1 - p) for _ in range(2)]).cpu().double()
    assert torch.equal(s, want)
    leaves = {n: v.double().requires_grad_(True) for n, v in P.items()}
    xd = x.double().requires_grad_(True)
    sub = lambda pfx: {k[len(pfx):]: v for k, v in leaves.items() if k.startswith(pfx)}      # noqa: E731
    x1 = R.frfn(F.layer_norm(xd, (dim,), leaves["norm1.weight"], leaves["norm1.bias"], 1e-5), sub("mlp."))
    f = torch.fft.ifftn(R.frfn(torch.fft.fftn(x1, dim=(-2, -1)).real, sub("freq_mlp.")), dim=(-2, -1)).real
    y64 = xd + s[0].view(B, 1, 1) * f + s[1].view(B, 1, 1) * x1
    (y64 * probe.double()).sum().backward()
    assert _rel(y, y64) < 2e-5 and _rel(xg.grad, xd.grad) < 5e-5
    for n, prm in m.named_parameters():
        if prm.grad is not None:
            assert _rel(prm.grad, leaves[n].grad) < 3e-4, n
