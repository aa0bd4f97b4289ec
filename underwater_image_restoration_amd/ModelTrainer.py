"""``ModelTrainer`` with the reference's call surface and step semantics
(src/ModelTrainer.py:26-204): ``ModelTrainer(in, ref, test_in, test_ref).train(args, arch,
num_epochs, learning_rate, device)``.

Per batch the reference does zero_grad -> forward -> loss -> backward -> clip_grad_norm_(1.0) ->
Adam/AdamW.step (:78-88, :197-204) with MultiStepLR([1,100,250], 0.25) per epoch (:55,129), then validates
with loss / PSNR / SSIM (:134-158) and saves ``{'epoch','model_state_dict','optimizer_state_dict','loss'}``
(:176-181).  Here the whole batch step is ``trainer.TrainStep`` (HIP launch lists + fused
clip/Adam, no host sync inside); the loop, the print format, the scheduler, the checkpoint keys,
the file names and the ``optimizer_state_dict`` layout (``torch.optim.Adam.state_dict()``) are the reference's.

What ``args`` is read for is exactly what the reference reads (``train_batch_size``, ``use_dwt``, ``lossf``, ``optim``,
``lr``, ``arch``); like the reference, ``--height/--width/--workers`` are NOT read: images are resized to 1024 x 1024
(src/DataManipulation/DataLoader.py:14) unless the caller sets the build's own ``args.image_size``.

Differences, all documented in DESIGN.md: W&B is not contacted (network service), the post-save visualisation of a
hard-coded file is skipped, ``device`` must be the GPU, validation PSNR / SSIM are printed (the reference only sends
them to W&B).  Under ``torch.distributed.run`` each process drives the GPU ``LOCAL_RANK`` names (one process per GPU),
reads its own shard of the data, and validation means are averaged over the ranks.
"""
from __future__ import annotations

import os
import time

import torch
import torch.distributed as dist

from . import Models
from . import parallel
from .DataManipulation.DataLoader import get_dataloaders
from .Losses.losses import LossFunction  # noqa: F401  (re-exported surface)
from .trainer import TrainStep, make_train_step  # noqa: F401

REFERENCE_IMAGE_SIZE = 1024      # transforms.Resize((1024, 1024)) hard-coded at src/DataManipulation/DataLoader.py:14
ADAMW_WEIGHT_DECAY = 0.01        # torch.optim.AdamW default; the reference passes only lr (src/ModelTrainer.py:201)


def multistep_lr(base_lr: float, epoch: int, milestones=(1, 100, 250), gamma: float = 0.25) -> float:
    return base_lr * (gamma ** sum(epoch >= m for m in milestones))


def resolve_device(device) -> torch.device:
    """'cuda' under torchrun means THIS rank's GPU (LOCAL_RANK); an explicit index is kept.  Sets the current device:
    every launch goes to torch.cuda.current_stream(), which must belong to the device that owns the buffers."""
    if not str(device).startswith("cuda"):
        raise RuntimeError("the MI355X path needs the GPU ('cuda' device in PyTorch-ROCm); there is no CPU fallback")
    dev = torch.device(device)
    if dev.index is None:
        dev = torch.device("cuda", int(os.environ.get("LOCAL_RANK", torch.cuda.current_device() if torch.cuda.is_available() else 0)))
    torch.cuda.set_device(dev)
    return dev


def optimizer_settings(args) -> float:
    """-> weight decay of the fused step for ``args.optim`` (src/ModelTrainer.py:197-204: 'adam' | 'adamw', else ValueError)."""
    name = getattr(args, "optim", "adam")
    if name == "adam":
        return 0.0
    if name == "adamw":
        return ADAMW_WEIGHT_DECAY
    raise ValueError(f"Unsupported optimizer: {name}")


def run_config(args, learning_rate: float) -> dict:
    """What the reference's ``train`` derives from ``args`` (src/ModelTrainer.py:37-46,60,197-204), and nothing else:
    batch size, arch flag ``use_dwt``, loss name, optimizer; the image size is the loader's constant 1024
    (DataLoader.py:14) - ``args.height / args.width`` (defaults 128 / 256, args.py:42-43) are parsed and ignored there,
    so they are ignored here; only the build's own ``args.image_size`` overrides it.  Loader workers: the reference calls
    ``get_dataloaders`` without ``num_workers`` (default 4; ``--workers`` is ignored, :38)."""
    return {"batch_size": args.train_batch_size,
            "image_size": getattr(args, "image_size", None) or REFERENCE_IMAGE_SIZE,
            "num_workers": getattr(args, "loader_workers", 4),
            "device_resize": bool(getattr(args, "device_resize", True)),
            "use_dwt": getattr(args, "use_dwt", "Fourier"),
            "lossf": args.lossf,
            "lr": learning_rate,
            "weight_decay": optimizer_settings(args)}


class ModelTrainer:
    def __init__(self, inputDirectory, referenceDirectory, testInputDirectory, testReferenceDirectory):
        self.inputDir = inputDirectory
        self.referenceDir = referenceDirectory
        self.testInputDir = testInputDirectory
        self.testReferenceDir = testReferenceDirectory

    def train(self, args, arch="SpectralTransformer", num_epochs=10, learning_rate=0.001,
              device="cuda" if torch.cuda.is_available() else "cpu"):
        dev = resolve_device(device)
        parallel.init_from_env(dev)
        world = parallel.world_size()
        rank0 = parallel.rank() == 0
        cfg = run_config(args, learning_rate)
        image_size = cfg["image_size"]
        print("Preparing data loaders...batch size" + str(args.train_batch_size))
        train_loader, test_loader = get_dataloaders(self.inputDir, self.referenceDir, self.testInputDir, self.testReferenceDir,
                                                    cfg["batch_size"], num_workers=cfg["num_workers"], image_size=image_size,
                                                    device_resize=cfg["device_resize"], device=dev)
        print("Initializing model...")
        model = Models.init_model(name=arch, use_dwt=cfg["use_dwt"]).to(dev)
        step = make_train_step(model, cfg["lossf"], learning_rate, cfg["weight_decay"])
        step.check_memory(cfg["batch_size"], image_size, image_size)
        t0 = time.time()
        directory = f"{args.lossf}-{args.lr}-{args.arch}-{t0}-{getattr(args, 'use_dwt', 'Fourier')}//"
        best_loss = float("inf")
        print(f"Starting training for {num_epochs} epochs...")
        for epoch in range(num_epochs):
            model.train()
            sampler = getattr(getattr(train_loader, "loader", train_loader), "sampler", None)
            if hasattr(sampler, "set_epoch"):
                sampler.set_epoch(epoch)             # DistributedSampler: a new shuffle every epoch
            lr = multistep_lr(learning_rate, epoch)
            epoch_loss, start = 0.0, time.time()
            pending = []
            for batch, (raw_imgs, ref_imgs) in enumerate(train_loader):
                raw_imgs = raw_imgs.to(dev, non_blocking=True)
                ref_imgs = ref_imgs.to(dev, non_blocking=True)
                loss, norm = step(raw_imgs, ref_imgs, lr=lr)
                pending.append((batch, loss.clone(), norm.clone()))
                if len(pending) >= 8 or batch + 1 == len(train_loader):   # read back in bulk: no per-batch sync
                    for b, l, n in pending:
                        lv = float(l)
                        epoch_loss += lv
                        print(f"Batch {b + 1}/{len(train_loader)}, Loss: {lv:.6f}, Norm: {float(n):.6f}")
                    pending.clear()
            epoch_time = time.time() - start
            avg = epoch_loss / max(1, len(train_loader))
            print(f"Epoch {epoch + 1}/{num_epochs} completed in {epoch_time:.2f}s, Avg Loss: {avg:.6f}")
            model.eval()
            avg_val_loss, avg_psnr, avg_ssim = self.validate(step, test_loader, dev, world)
            print(f"Validation Loss: {avg_val_loss:.6f}")
            print(f"Validation PSNR: {avg_psnr:.4f}")
            print(f"Validation SSIM: {avg_ssim:.4f}")
            if rank0:
                best_loss = self.SaveModel(avg_val_loss, best_loss, directory, epoch, model, step.opt)
        print("Training completed!")
        return model

    @staticmethod
    def validate(step, test_loader, dev, world=1):
        """Per-batch loss / PSNR / SSIM means (src/ModelTrainer.py:134-158).  With several ranks each one evaluates the
        batches ``rank, rank + world, ...`` and the sums are all-reduced (three scalars + a count).  get_dataloaders hands every
        rank a loader of ITS batches only (RankBatches: nothing is decoded for the others); any other iterable is strided here."""
        rank = parallel.rank()
        acc = torch.zeros(4, dtype=torch.float64, device=dev)
        inner = getattr(test_loader, "loader", test_loader)
        sharded = getattr(getattr(inner, "batch_sampler", None), "rank_sharded", False)
        for i, (raw_imgs, ref_imgs) in enumerate(test_loader):
            if world > 1 and not sharded and i % world != rank:
                continue
            vl, vp, vs = step.evaluate(raw_imgs.to(dev), ref_imgs.to(dev), with_ssim=True)   # :148-150
            acc += torch.stack([vl.double(), vp.double(), vs.double(), torch.ones((), dtype=torch.float64, device=dev)])
        if world > 1:
            dist.all_reduce(acc, op=dist.ReduceOp.SUM)
        a = acc.tolist()
        n = max(1.0, a[3])
        return a[0] / n, a[1] / n, a[2] / n

    def SaveModel(self, avg_val_loss, best_loss, directory, epoch, model, optimizer):
        os.makedirs(directory, exist_ok=True)
        best = avg_val_loss < best_loss
        name = f"best_spectral_transformer_{epoch}.pth" if best else "latest_spectroformer.pth"

        def cpu(o):
            if torch.is_tensor(o):
                return o.detach().cpu()
            if isinstance(o, dict):
                return {k: cpu(v) for k, v in o.items()}
            if isinstance(o, (list, tuple)):
                return type(o)(cpu(v) for v in o)
            return o
        torch.save({"epoch": epoch,
                    "model_state_dict": cpu(dict(model.state_dict())),
                    "optimizer_state_dict": cpu(optimizer.state_dict()),
                    "loss": avg_val_loss}, directory + name)
        if best:
            print(f"Model saved with loss: {avg_val_loss:.6f}")
        return min(best_loss, avg_val_loss)
