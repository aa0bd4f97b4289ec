"""Timing-only experiment (WRONG results, never a product path): the headline step with some recorded launches replaced by no-ops.
usage: python tools/ab/skip.py [--steps 10] [--warmup 3] SPEC [SPEC ...]
SPEC = label:lane=<0|1|4>   every launch of that lane      | label:name=<regex>   launches whose op name matches
       label:none           the unmodified step (baseline arm)
Arms run in the order given, `--rounds` times, inside ONE process on one GPU; prints ms/step per arm and round."""
import argparse
import os
import re
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--rounds", type=int, default=2)
    ap.add_argument("specs", nargs="+")
    a = ap.parse_args()
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    from underwater_image_restoration_amd import init_model
    from underwater_image_restoration_amd.trainer import TrainStep
    torch.manual_seed(1234)
    model = init_model("SpectralTransformer").to(dev)
    step = TrainStep(model, "L1withColor", lr=3e-4)
    g = torch.Generator(device="cpu").manual_seed(1000)
    raw = (torch.rand(16, 3, 256, 256, generator=g) * 2 - 1).to(dev)
    ref = (torch.rand(16, 3, 256, 256, generator=g) * 2 - 1).to(dev)
    for _ in range(3):
        step(raw, ref)
    torch.cuda.synchronize()
    plan = model._plan_for(raw)
    orig = {id(l): list(l) for l in (plan.fwd, plan.bwd)}

    def noop(*args):
        return 0

    def apply(spec):
        label, _, rule = spec.partition(":")
        n = 0
        for l in (plan.fwd, plan.bwd):
            l[:] = orig[id(l)]
            if rule in ("", "none"):
                continue
            key, _, val = rule.partition("=")
            lanes = plan.lanes[id(l)]
            for i, (fn, args, name) in enumerate(l):
                if fn is None:
                    continue
                hit = (lanes[i] == int(val)) if key == "lane" else bool(re.search(val, name))
                if hit:
                    l[i] = (noop, args, name)
                    n += 1
        return label, n

    for r in range(a.rounds):
        for spec in a.specs:
            label, n = apply(spec)
            for _ in range(a.warmup):
                step(raw, ref)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(a.steps):
                step(raw, ref)
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / a.steps * 1e3
            print(f"round {r} {label:28s} skipped {n:4d} launches  {ms:8.3f} ms/step", flush=True)
    apply("x:none")


if __name__ == "__main__":
    main()
