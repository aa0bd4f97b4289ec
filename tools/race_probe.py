"""Which gradient slots differ between the serial schedule and the overlapped 3-stream schedule?"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from underwater_image_restoration_amd import init_model, ops, _lib as L
torch.manual_seed(7)
model = init_model("SpectralTransformer").to("cuda")
fp = model.flat_params()
B, S = 16, 256
g = torch.Generator(device="cpu").manual_seed(3)
raw = (torch.rand(B, 3, S, S, generator=g) * 2 - 1).cuda()
ref = (torch.rand(B, 3, S, S, generator=g) * 2 - 1).cuda()
plan = model._plan_for(raw)
def run(serial):
    out = plan.forward(raw, serial=serial).clone()
    dout = ops.loss_grad(out, ref, torch.ones(1, device="cuda"), L.LOSS_KINDS["L1withColor"])
    fp.grad[: fp.n_live].fill_(float("nan"))
    plan.backward(dout, serial=serial)
    torch.cuda.synchronize()
    return out, fp.grad.clone()
o0, g0 = run(True)
for trial in range(3):
    o1, g1 = run(False)
    print("trial", trial, "out equal", torch.equal(o0, o1), "grad equal", torch.equal(g0, g1), "nan in serial", int(torch.isnan(g0).sum()), "nan in overlapped", int(torch.isnan(g1).sum()))
    bad = []
    for n in fp.live_names:
        o, k = fp.slots[n]
        a, b = g0[o:o + k], g1[o:o + k]
        if not torch.equal(a, b) and not (torch.isnan(a) & torch.isnan(b)).all():
            bad.append((n, float((a - b).abs().max())))
    print("   differing tensors:", len(bad), bad[:12])
