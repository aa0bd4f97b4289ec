"""Train step fed from image files: images/s of the SpectralTransformer step at B=16, 256x256 when
the batch comes (a) from HBM-resident tensors (the bench.py workload), (b) from the CPU loader of the reference
(PIL decode + PIL resize + ToTensor + Normalize in the workers, fp32 batch over PCIe), (c) from the device-resize loader
(workers decode only, uint8 over PCIe, one resize launch per batch, prefetched one batch ahead).
usage: python tools/bench_pipeline.py [--workers 8] [--epochs 3] [--files 128]   -> one JSON line"""
import argparse, json, os, sys, tempfile, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from underwater_image_restoration_amd import Models                                    # noqa: E402
from underwater_image_restoration_amd.trainer import TrainStep                         # noqa: E402
from underwater_image_restoration_amd.DataManipulation.DataLoader import get_dataloaders  # noqa: E402


def make_files(root, n):
    from PIL import Image
    rng = np.random.default_rng(0)
    sizes = [(480, 640), (375, 500), (600, 800), (768, 1024)]
    for d in ("raw", "ref"):
        os.makedirs(os.path.join(root, d))
        for i in range(n):
            h, w = sizes[i % len(sizes)]
            yy, xx = np.mgrid[0:h, 0:w]
            base = (np.sin(xx / (20 + i % 7)) + np.cos(yy / (15 + i % 5)))[..., None] * 60 + 128
            img = np.clip(base + rng.normal(0, 12, (h, w, 3)), 0, 255).astype(np.uint8)   # smooth + noise: PNG-realistic
            Image.fromarray(img).save(os.path.join(root, d, f"{i}_img_.png"))


HOST = {}


def run(step, loader, epochs, dev, tag=None):
    """images/s; with `tag`, also the host time per batch spent waiting for the loader and enqueuing the step."""
    n = nb = 0
    t_load = t_step = 0.0
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(epochs):
        it = iter(loader)
        while True:
            ta = time.perf_counter()
            try:
                raw, ref = next(it)
            except StopIteration:
                break
            raw = raw.to(dev, non_blocking=True)
            ref = ref.to(dev, non_blocking=True)
            tb = time.perf_counter()
            step(raw, ref, lr=3e-4)
            tc = time.perf_counter()
            t_load += tb - ta
            t_step += tc - tb
            n += raw.shape[0]
            nb += 1
    torch.cuda.synchronize()
    if tag:
        HOST[tag] = {"host_ms_per_batch_in_loader": 1e3 * t_load / nb, "host_ms_per_batch_enqueuing_step": 1e3 * t_step / nb}
    return n / (time.perf_counter() - t0)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workers", type=int, default=8)
    ap.add_argument("--epochs", type=int, default=3)
    ap.add_argument("--files", type=int, default=128)
    args = ap.parse_args()
    dev = torch.device("cuda")
    torch.manual_seed(0)
    model = Models.init_model("SpectralTransformer").to(dev)
    step = TrainStep(model, "L1", lr=3e-4)
    out = {"workload": "SpectralTransformer train step, B=16, 256x256, fp32; files: PNG 375x500..768x1024", "workers": args.workers}
    g = torch.Generator().manual_seed(1)
    raw = (torch.rand(16, 3, 256, 256, generator=g) * 2 - 1).to(dev)
    ref = (torch.rand(16, 3, 256, 256, generator=g) * 2 - 1).to(dev)
    for _ in range(5):
        step(raw, ref, lr=3e-4)
    out["resident_images_per_s"] = run(step, [(raw, ref)] * 24, 1, dev)
    with tempfile.TemporaryDirectory() as root:
        make_files(root, args.files)
        a, b = os.path.join(root, "raw"), os.path.join(root, "ref")
        for name, kw in (("cpu_loader_images_per_s", {}), ("device_resize_loader_images_per_s", {"device_resize": True, "device": dev})):
            tr, _ = get_dataloaders(a, b, a, b, batch_size=16, num_workers=args.workers, image_size=256, **kw)
            run(step, tr, 1, dev)                       # warm-up epoch (worker start-up, staging buffers)
            out[name] = run(step, tr, args.epochs, dev, tag=name)
        # what the loaders alone deliver (no train step): the host-side ceiling of each
        for name, kw in (("cpu_loader_alone_images_per_s", {}), ("device_resize_loader_alone_images_per_s", {"device_resize": True, "device": dev})):
            tr, _ = get_dataloaders(a, b, a, b, batch_size=16, num_workers=args.workers, image_size=256, **kw)
            for _ in tr:
                pass
            torch.cuda.synchronize()
            t0, n = time.perf_counter(), 0
            for _ in range(args.epochs):
                for r_, f_ in tr:
                    n += r_.shape[0]
                    last = r_.to(dev, non_blocking=True)
            torch.cuda.synchronize()
            out[name] = n / (time.perf_counter() - t0)
    out["host_time"] = HOST
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
