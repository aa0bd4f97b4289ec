"""Measure UIQM on the GPU (reference uqim_utils.py:176-190): a batch of 256 x 256 images, the size
eval.ipynb evaluates at.  ONE JSON line:

  value        images/s of `metrics.uiqm` with the images resident in HBM (uint8 HWC, what cv2.imread + resize gives)
  roofline     algorithmic bytes (every image byte read once + 32 B of results) / launch-set time vs 8 TB/s - the path is NOT
               HBM-bound (ten L2-resident passes of radix select per image plane), the fraction says so
  cpu_baseline the oracle (vectorised numpy restatement, kind "port") on one host thread, bounded sample; the reference's own
               Python-loop implementation measured in the build container is quoted in DESIGN.md (it cannot travel)
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from underwater_image_restoration_amd import metrics  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu", action="store_true")
    args = ap.parse_args()
    rng = np.random.default_rng(0)
    N, S = args.batch, args.size
    imgs = rng.integers(0, 256, (N, S, S, 3), dtype=np.uint8)
    dev = torch.from_numpy(imgs).cuda()
    for _ in range(args.warmup):
        out = metrics.uiqm(dev)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(args.steps):
        out = metrics.uiqm(dev)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / args.steps
    alg = N * (S * S * 3 + 32)
    line = {"metric": "UIQM images/s (256x256 uint8, batch resident in HBM)", "value": N / (ms * 1e-3), "unit": "images/s",
            "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms, "higher_is_better": True,
            "dtype": "f32/f64 (the reference's own mix)", "data": "synthetic",
            "config": {"workload": f"getUIQM on {N} uint8 images of {S}x{S}x3"},
            "roofline": {"bound": "hbm", "achieved": alg / (ms * 1e-3) / 1e9, "peak": 8000.0, "unit": "GB/s",
                         "frac": alg / (ms * 1e-3) / 8e12, "traffic": None,
                         "note": "latency / L2-bound: 10 passes of radix select + variance per image plane"}}
    if not args.no_cpu:
        from oracle import uiqm_ref as U
        t0 = time.perf_counter()
        k = 0
        while time.perf_counter() - t0 < 10.0 and k < N:
            want = U.get_uiqm(imgs[k])
            got = out[k].cpu().numpy()
            assert all(abs(a - b) <= 1e-9 * max(1.0, abs(b)) for a, b in zip(got, want)), (k, got, want)
            k += 1
        dt = (time.perf_counter() - t0) / k
        line["cpu_baseline"] = {"value": 1.0 / dt, "unit": "images/s", "cores": 1, "kind": "port",
                                "sample": f"{k} images of the same batch through oracle/uiqm_ref.py (numpy), each checked against the GPU result"}
    print(json.dumps(line))


if __name__ == "__main__":
    main()
