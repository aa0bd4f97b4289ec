"""Round 3: the module networks at the sizes their benches time - NewBigFRFNModel 256 x 256 bs 8 (BASELINE
configs[3]), AST 512 x 512 bs 4 (configs[4]) - through size-independent properties: finite, bitwise run-to-run, batch
equivariance, micro-batched step == full-batch step (eval mode: stochastic depth off), and the memory model ModuleTrainStep
sizes micro-batches with."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
if not torch.cuda.is_available():  # pragma: no cover
    pytest.skip("needs a GPU", allow_module_level=True)

from underwater_image_restoration_amd import init_model                    # noqa: E402
from underwater_image_restoration_amd.trainer import make_train_step       # noqa: E402


def _fresh(arch, micro_batch=None):
    torch.manual_seed(1234)
    m = init_model(arch).cuda()
    m.eval()                                   # DropPath factors are exactly 1: the step is deterministic
    step = make_train_step(m, "charbonnier", 3e-4)
    step.micro_batch = micro_batch
    return m, step


@pytest.mark.parametrize("arch,B,S", [("NewBigFRFNModel", 8, 256), ("AST", 4, 512)])
def test_full_size_properties(arch, B, S):
    g = torch.Generator().manual_seed(7)
    x = (torch.rand(B, 3, S, S, generator=g) * 2 - 1).cuda()
    y = (torch.rand(B, 3, S, S, generator=g) * 2 - 1).cuda()
    m, step = _fresh(arch)
    with torch.no_grad():
        o1 = m(x).clone()
        o2 = m(x).clone()
        perm = torch.arange(B - 1, -1, -1, device="cuda")
        o3 = m(x[perm].contiguous())
    assert torch.isfinite(o1).all() and torch.equal(o1, o2), "forward must be finite and bitwise run-to-run"
    assert float((o3 - o1[perm]).abs().max()) <= 1e-6 * float(o1.abs().max()), "samples are independent: batch order must not matter"
    for mod in step.mods:                      # forward-only plans of the checks above: the memory figure below is the train plans'
        mod._plans.clear()
    loss_full, norm_full = step(x, y)
    torch.cuda.synchronize()
    g_full = step.bucket.clone()
    assert np.isfinite(float(loss_full)) and np.isfinite(float(norm_full)) and torch.isfinite(g_full).all()
    # memory model (ModuleTrainStep.check_memory): an upper estimate, within 12 % + the fixed part
    real, est = step.plan_bytes(), step.plan_bytes_estimate(B, S, S)
    print(f"{arch} {B} x {S}^2: plans {real / 2**30:.1f} GiB, estimate {est / 2**30:.1f} GiB")
    assert real <= est <= 1.12 * real + (1 << 30), (real, est)
    del m, step
    torch.cuda.empty_cache()
    # the same step as two micro-batches with gradient accumulation (what check_memory falls back to when the plans do not fit)
    m2, step2 = _fresh(arch, micro_batch=B // 2)
    loss_mb, norm_mb = step2(x, y)
    torch.cuda.synchronize()
    g_mb = step2.bucket
    assert abs(float(loss_mb) - float(loss_full)) <= 2e-6 * abs(float(loss_full))
    assert abs(float(norm_mb) - float(norm_full)) <= 2e-5 * abs(float(norm_full))
    assert float((g_mb - g_full).abs().max()) <= 5e-5 * float(g_full.abs().max())
    # evaluation recombines the same way
    with torch.no_grad():
        l2, p2 = step2.evaluate(x, y)
        step2.micro_batch = B
        l1, p1 = step2.evaluate(x, y)
    assert abs(float(l1) - float(l2)) <= 2e-6 * abs(float(l1)) and abs(float(p1) - float(p2)) <= 1e-4


def test_module_step_refuses_what_cannot_fit_with_the_numbers():
    """AST at the reference's default 1024 x 1024 (DataLoader.py:14) keeps ~67 GB of plans per image: a batch of 8 must come back as
    micro-batches from check_memory (not as an allocator OOM deep inside plan recording), and an impossible size must raise up front."""
    m, step = _fresh("AST")
    mb = step.check_memory(8, 1024, 1024)
    assert 1 <= mb < 8 and 8 % mb == 0
    with pytest.raises(RuntimeError, match="GiB"):
        step.check_memory(1, 4096, 4096)


def test_new_shapes_are_sized_against_free_memory_not_against_plans_that_stay_alive(monkeypatch):
    """the modules keep the plans of other shapes alive, so a validation tail batch sized against "free + existing plans"
    ran out of memory while recording its plans beside training plans that fill the HBM.  With the device's free memory pinned to a
    budget just above one training micro-batch: (1) validation of the training shape reuses the training micro-batch (no second set
    of plans), (2) a tail batch that cannot get its own plans beside the training plans makes the step drop them FIRST and is then
    sized against what is really free - and both still evaluate correctly."""
    from underwater_image_restoration_amd import engine
    m, step = _fresh("NewBigFRFNModel")
    S, B = 128, 4
    x = torch.rand(B, 3, S, S, device="cuda")
    y = torch.rand(B, 3, S, S, device="cuda")
    need2 = step.plan_bytes_estimate(2, S, S)
    budget = int(need2 / 0.92 * 1.2)                       # room for plans of micro-batch 2 (not 4), once

    def fake_free(dev):
        return max(0, budget - step.plan_bytes())
    monkeypatch.setattr(engine, "device_bytes_available", fake_free)
    assert step.micro_batch_for(B, S, S) == 2
    step(x, y)
    torch.cuda.synchronize()
    held = step.plan_bytes()
    assert held > 0
    # (1) same shape, validation: the training micro-batch and its plans
    with torch.no_grad():
        l_a, p_a = step.evaluate(x, y)
    assert step.micro_batch_for(B, S, S, train=False) == 2 and step.plan_bytes() == held
    # (2) a tail batch of 3 images (micro-batch 3 or 1): its forward-only plans do not fit beside the training plans
    with torch.no_grad():
        l_t, p_t = step.evaluate(x[:3], y[:3])
        torch.cuda.synchronize()
    assert np_isfinite(float(l_t)) and np_isfinite(float(p_t))
    # the training shape comes back afterwards and trains on
    l2, n2 = step(x, y)
    assert np_isfinite(float(l2)) and np_isfinite(float(n2))


def np_isfinite(v):
    import math
    return math.isfinite(v)
