"""``ModelTrainer`` with the reference's call surface and step semantics
(src/ModelTrainer.py:26-204): ``ModelTrainer(in, ref, test_in, test_ref).train(args, arch,
num_epochs, learning_rate, device)``.

Per batch the reference does zero_grad -> forward -> loss -> backward -> clip_grad_norm_(1.0) ->
Adam.step (:78-88) with MultiStepLR([1,100,250], 0.25) per epoch (:55,129), then validates with
loss / PSNR (:134-158) and saves ``{'epoch','model_state_dict','optimizer_state_dict','loss'}``
(:176-181).  Here the whole batch step is ``trainer.TrainStep`` (HIP launch lists + fused
clip/Adam, no host sync inside); the loop, the print format, the scheduler and the checkpoint keys
and file names are the reference's.  Differences, all documented in DESIGN.md: W&B is not
contacted (network service), SSIM is not computed (third-party pytorch_msssim), the post-save
visualisation of a hard-coded file is skipped, ``device`` must be the GPU.
"""
from __future__ import annotations

import os
import time

import torch

from . import Models
from . import parallel
from .DataManipulation.DataLoader import get_dataloaders
from .Losses.losses import LossFunction  # noqa: F401  (re-exported surface)
from .trainer import TrainStep


def multistep_lr(base_lr: float, epoch: int, milestones=(1, 100, 250), gamma: float = 0.25) -> float:
    return base_lr * (gamma ** sum(epoch >= m for m in milestones))


class ModelTrainer:
    def __init__(self, inputDirectory, referenceDirectory, testInputDirectory, testReferenceDirectory):
        self.inputDir = inputDirectory
        self.referenceDir = referenceDirectory
        self.testInputDir = testInputDirectory
        self.testReferenceDir = testReferenceDirectory

    def train(self, args, arch="SpectralTransformer", num_epochs=10, learning_rate=0.001,
              device="cuda" if torch.cuda.is_available() else "cpu"):
        if not str(device).startswith("cuda"):
            raise RuntimeError("the MI355X path needs the GPU ('cuda' device in PyTorch-ROCm); there is no CPU fallback")
        dev = torch.device(device)
        parallel.init_from_env(dev)
        rank0 = parallel.rank() == 0
        image_size = getattr(args, "image_size", None) or getattr(args, "height", None) or 1024
        print("Preparing data loaders...batch size" + str(args.train_batch_size))
        train_loader, test_loader = get_dataloaders(self.inputDir, self.referenceDir, self.testInputDir, self.testReferenceDir,
                                                    args.train_batch_size, num_workers=getattr(args, "workers", 4),
                                                    image_size=image_size,
                                                    device_resize=bool(getattr(args, "device_resize", True)), device=dev)
        print("Initializing model...")
        model = Models.init_model(name=arch, use_dwt=getattr(args, "use_dwt", "Fourier")).to(dev)
        if getattr(args, "optim", "adam") != "adam":
            raise ValueError(f"Unsupported optimizer: {args.optim} (the fused step implements the reference default, Adam)")
        step = TrainStep(model, args.lossf, lr=learning_rate)
        t0 = time.time()
        directory = f"{args.lossf}-{args.lr}-{args.arch}-{t0}-{getattr(args, 'use_dwt', 'Fourier')}//"
        best_loss = float("inf")
        print(f"Starting training for {num_epochs} epochs...")
        for epoch in range(num_epochs):
            model.train()
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
            val_loss = psnr = ssim = 0.0
            for raw_imgs, ref_imgs in test_loader:
                vl, vp, vs = step.evaluate(raw_imgs.to(dev), ref_imgs.to(dev), with_ssim=True)   # :148-150
                val_loss += float(vl)
                psnr += float(vp)
                ssim += float(vs)
            avg_val_loss = val_loss / max(1, len(test_loader))
            print(f"Validation Loss: {avg_val_loss:.6f}")
            print(f"Validation PSNR: {psnr / max(1, len(test_loader)):.4f}")
            print(f"Validation SSIM: {ssim / max(1, len(test_loader)):.4f}")
            if rank0:
                best_loss = self.SaveModel(avg_val_loss, best_loss, directory, epoch, model, step.opt)
        print("Training completed!")
        return model

    def SaveModel(self, avg_val_loss, best_loss, directory, epoch, model, optimizer):
        os.makedirs(directory, exist_ok=True)
        best = avg_val_loss < best_loss
        name = f"best_spectral_transformer_{epoch}.pth" if best else "latest_spectroformer.pth"
        torch.save({"epoch": epoch,
                    "model_state_dict": {k: v.detach().cpu() for k, v in model.state_dict().items()},
                    "optimizer_state_dict": {k: (v.detach().cpu() if torch.is_tensor(v) else v)
                                             for k, v in optimizer.state_dict().items()},
                    "loss": avg_val_loss}, directory + name)
        if best:
            print(f"Model saved with loss: {avg_val_loss:.6f}")
        return min(best_loss, avg_val_loss)
