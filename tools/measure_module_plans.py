"""Plan memory per input pixel of the module networks (PLAN_BYTES_PER_PIXEL_* of AST / MyBigFRFNModel): build, run one step, sum plan bytes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from underwater_image_restoration_amd import init_model
from underwater_image_restoration_amd.trainer import make_train_step
for arch, shapes in (("AST", [(1, 128), (2, 256), (1, 512)]), ("NewBigFRFNModel", [(1, 128), (2, 128), (2, 256)])):
    for B, S in shapes:
        torch.manual_seed(0)
        m = init_model(arch).cuda()
        step = make_train_step(m, "charbonnier", 3e-4)
        step.micro_batch = B
        x, y = torch.rand(B, 3, S, S, device="cuda") * 2 - 1, torch.rand(B, 3, S, S, device="cuda") * 2 - 1
        step(x, y); torch.cuda.synchronize()
        tb = step.plan_bytes()
        for mod in step.mods:
            mod._plans.clear()
        m.eval(); step.evaluate(x, y); torch.cuda.synchronize()
        eb = step.plan_bytes()
        print(f"{arch} B={B} {S}x{S}: train {tb / (B * S * S):9.0f} B/px ({tb / 2**30:.2f} GiB)  eval {eb / (B * S * S):9.0f} B/px", flush=True)
        del m, step
        torch.cuda.empty_cache()
