"""oracle/ast_ref.py against the outputs and gradients of the reference's own AST classes (tests/golden/ast_ref.npz, recorded by
tests/golden/make_ast_vectors.py from /root/reference/src/Models/AST.py).  CPU only."""
import ast
import os

import numpy as np
import pytest
import torch

from oracle import ast_ref as O

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "ast_ref.npz"))
S = 16


def _params(tag, salt, dtype=torch.float64):
    shapes = {n: ast.literal_eval(s) for n, s in zip(G[tag + ".param_names"], G[tag + ".param_shapes"])}
    return {k: v.to(dtype) for k, v in O.closed_form_params(shapes, salt, torch.float32).items()}


def _close(a, b, tol, what):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    err = np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)
    assert err <= tol, f"{what}: {err:.3e} > {tol}"


CASES = {
    "blk_att_s0": (lambda x, P: O.transformer_block(x, P, "", 2, 0, True), "astblk0", (2, S * S, 64, 171), (2, S * S, 64, 172)),
    "blk_att_s4": (lambda x, P: O.transformer_block(x, P, "", 2, 4, True), "astblk4", (2, S * S, 64, 171), (2, S * S, 64, 172)),
    "blk_noatt": (lambda x, P: O.transformer_block(x, P, "", 1, 0, False), "astblkn", (2, S * S, 32, 173), (2, S * S, 32, 174)),
    "down_32": (lambda x, P: O.downsample(x, P), "astdown", (2, S * S, 32, 173), (2, (S // 2) ** 2, 64, 175)),
    "up_64": (lambda x, P: O.upsample(x, P), "astup", (2, S * S, 64, 171), (2, 4 * S * S, 32, 176)),
    "outproj": (lambda x, P: O.output_proj(x, P), "astout", (2, S * S, 64, 171), (2, 3, S * S, 178)),
}


@pytest.mark.parametrize("tag", list(CASES))
def test_block_oracle_matches_reference(tag):
    fn, salt, xin, pin = CASES[tag]
    P = _params(tag, salt)
    x = O.closed_form_input(*xin).double()
    probe = O.closed_form_input(*pin).double().reshape(G[tag + ".y"].shape)
    y, dxs, gr = O.grads(fn, [x], P, probe)
    _close(y, G[tag + ".y"], 2e-6, tag + ".y")
    _close(dxs[0], G[tag + ".dx0"], 2e-6, tag + ".dx")
    for n in G[tag + ".grad_names"]:
        _close(gr[str(n)], G[f"{tag}.grad.{n}"], 5e-6, f"{tag}.grad.{n}")
    assert len(G[tag + ".dead"]) == 0


def test_input_proj_oracle_matches_reference():
    P = _params("inproj", "astin")
    img = O.closed_form_input(2, 3, S * S, 177).reshape(2, 3, S, S).double()
    probe = O.closed_form_input(2, S * S, 32, 174).double()
    y, _, gr = O.grads(lambda x, Pp: O.input_proj(x, Pp), [img], P, probe)
    _close(y, G["inproj.y"], 2e-6, "inproj.y")
    for n in G["inproj.grad_names"]:
        _close(gr[str(n)], G[f"inproj.grad.{n}"], 5e-6, f"inproj.grad.{n}")


def test_whole_net_oracle_matches_reference():
    """1 x 3 x 128 x 128 through all nine layers (bottleneck = one cyclically shifted 8 x 8 window): output, every gradient norm,
    thirteen full gradients."""
    tag, N = "net128", 128
    P = _params(tag, "astnet")
    x = O.closed_form_input(1, 3, N * N, 179).reshape(1, 3, N, N).double()
    probe = O.closed_form_input(1, 3, N * N, 180).reshape(1, 3, N, N).double()
    y, _, gr = O.grads(lambda t, Pp: O.ast_net(t, Pp), [x], P, probe)
    _close(y, G[tag + ".y"], 2e-6, "net.y")
    names, norms = [str(n) for n in G[tag + ".grad_names"]], G[tag + ".grad_norms"]
    assert len(names) == 264 and len(G[tag + ".dead"]) == 0          # every parameter of AST is used
    for n, ref in zip(names, norms):
        got = float(gr[n].norm())
        assert abs(got - ref) <= 1e-5 * max(ref, 1e-12), (n, got, ref)
    for k in G.files:
        if k.startswith(tag + ".grad.") and k not in (tag + ".grad_names", tag + ".grad_norms"):
            _close(gr[k[len(tag) + 6:]], G[k], 1e-5, k)


def test_shift_mask_regions():
    m = O.shift_mask(16, 16, 4)
    assert m.shape == (4, 64, 64) and float(m[0].abs().max()) == 0.0           # the first window lies inside region 0
    assert set(np.unique(m.numpy())) == {-100.0, 0.0}
    assert float(m[3].min()) == -100.0 and float(m[3, 0, 0]) == 0.0
