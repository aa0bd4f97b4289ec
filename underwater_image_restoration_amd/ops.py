"""Direct (un-planned) calls into the C ABI for the step tail: losses, PSNR, clip+Adam.
All tensors are fp32 CUDA(HIP) tensors; every call is asynchronous on the current stream."""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib as L

lib = L.lib
_LOSS_BLOCKS = 512
_SUMSQ_BLOCKS = 256


def _stream() -> C.c_void_p:
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _req(t: torch.Tensor, what: str) -> torch.Tensor:
    if t.device.type != "cuda":
        raise RuntimeError(f"{what}: expected a GPU tensor (no CPU fallback in the HIP path)")
    if t.dtype != torch.float32:
        t = t.float()
    return t.contiguous()


def loss_sums(pred: torch.Tensor, truth: torch.Tensor, kind: int) -> torch.Tensor:
    """-> device tensor [2] = (loss(kind), PSNR)."""
    pred, truth = _req(pred, "loss"), _req(truth, "loss")
    if pred.shape != truth.shape or pred.dim() != 4 or pred.shape[1] != 3:
        raise ValueError(f"loss expects matching [B,3,H,W] tensors, got {tuple(pred.shape)} / {tuple(truth.shape)}")
    B, _, H, W = pred.shape
    slab = torch.empty(_LOSS_BLOCKS * 8, dtype=torch.float32, device=pred.device)
    sums = torch.empty(8, dtype=torch.float32, device=pred.device)
    out = torch.empty(2, dtype=torch.float32, device=pred.device)
    st = _stream()
    L.check(lib.uir_loss_partials(pred.data_ptr(), truth.data_ptr(), slab.data_ptr(), _LOSS_BLOCKS, B, H * W, st), "uir_loss_partials")
    L.check(lib.uir_reduce_slab(slab.data_ptr(), _LOSS_BLOCKS, 8, sums.data_ptr(), 8, 0, 1.0, st), "uir_reduce_slab")
    L.check(lib.uir_loss_finalize(sums.data_ptr(), out.data_ptr(), kind, B, H * W, st), "uir_loss_finalize")
    return out


def ssim(x: torch.Tensor, y: torch.Tensor, data_range: float = 1.0) -> torch.Tensor:
    """0-dim device tensor = pytorch_msssim.ssim(x, y, data_range=data_range, size_average=True), the reference's torchSSIM
    (src/ModelTrainer.py:23-24): 11-tap Gaussian (sigma 1.5), valid convolution, K = (0.01, 0.03).  [N,C,H,W] fp32, H, W >= 11."""
    x, y = _req(x, "ssim"), _req(y, "ssim")
    if x.shape != y.shape or x.dim() != 4:
        raise ValueError(f"ssim expects matching [N,C,H,W] tensors, got {tuple(x.shape)} / {tuple(y.shape)}")
    N, Cc, H, W = x.shape
    nb = lib.uir_ssim_nblocks(N * Cc, H, W)
    if nb < 0:
        raise ValueError("ssim needs H, W >= 11 (the Gaussian window)")
    coords = torch.arange(11, dtype=torch.float32) - 5            # _fspecial_gauss_1d(11, 1.5), fp32 as in the package
    g = torch.exp(-(coords ** 2) / (2 * 1.5 ** 2))
    g = g / g.sum()
    slab = torch.empty(nb, dtype=torch.float32, device=x.device)
    out = torch.empty(1, dtype=torch.float32, device=x.device)
    a = L.SsimArgs()
    a.X, a.Y, a.slab, a.slab_floats = x.data_ptr(), y.data_ptr(), slab.data_ptr(), nb
    a.planes, a.H, a.W = N * Cc, H, W
    a.C1, a.C2 = (0.01 * data_range) ** 2, (0.03 * data_range) ** 2
    for i in range(11):
        a.win[i] = float(g[i])
    st = _stream()
    L.check(lib.uir_ssim_partials(C.byref(a), st), "uir_ssim_partials")
    L.check(lib.uir_reduce_slab(slab.data_ptr(), nb, 1, out.data_ptr(), 1, 0, 1.0 / (N * Cc * (H - 10) * (W - 10)), st), "uir_reduce_slab")
    return out[0]


def loss_grad(pred: torch.Tensor, truth: torch.Tensor, gscale: torch.Tensor, kind: int) -> torch.Tensor:
    pred, truth = _req(pred, "loss"), _req(truth, "loss")
    B, _, H, W = pred.shape
    gs = _req(gscale.reshape(1), "loss")
    dpred = torch.empty_like(pred)
    L.check(lib.uir_loss_bwd(pred.data_ptr(), truth.data_ptr(), gs.data_ptr(), dpred.data_ptr(), kind, B, H * W, _stream()), "uir_loss_bwd")
    return dpred


class FusedClipAdam:
    """clip_grad_norm_(max_norm) + Adam(lr, betas=(0.9,0.999), eps=1e-8, no weight decay) - the step
    tail of src/ModelTrainer.py:87-88,199 - as two kernels over the flat live-parameter buffer."""

    def __init__(self, flat, lr: float = 3e-4, betas=(0.9, 0.999), eps: float = 1e-8, max_norm: float = 1.0):
        self.flat = flat
        self.n = flat.n_live
        self.lr, self.betas, self.eps, self.max_norm = lr, betas, eps, max_norm
        dev = flat.device
        self.m = torch.zeros(self.n, dtype=torch.float32, device=dev)
        self.v = torch.zeros(self.n, dtype=torch.float32, device=dev)
        self.slab = torch.empty(_SUMSQ_BLOCKS, dtype=torch.float32, device=dev)
        self.sumsq = torch.zeros(1, dtype=torch.float32, device=dev)
        self.norm = torch.zeros(1, dtype=torch.float32, device=dev)
        self.t = 0
        self.args = L.AdamArgs()
        # per-step scalars {lr, bc1, bc2, gscale} live in device memory (written from a pinned host
        # staging buffer on the launch stream), so the kernels can be captured once in a hipGraph
        self.hyper = torch.zeros(4, dtype=torch.float32, device=dev)
        self._hyper_host = torch.zeros(4, dtype=torch.float32).pin_memory() if dev.type == "cuda" else torch.zeros(4)

    def advance(self, lr: float | None = None, grad_scale: float = 1.0) -> None:
        """Host side of one step: bump the step count and stage this step's scalars (async H2D, 16 bytes)."""
        if lr is not None:
            self.lr = lr
        self.t += 1
        h = self._hyper_host
        h[0], h[1], h[2], h[3] = self.lr, 1.0 - self.betas[0] ** self.t, 1.0 - self.betas[1] ** self.t, grad_scale
        self.hyper.copy_(h, non_blocking=True)

    def launch(self) -> torch.Tensor:
        """Device side: sum of squares -> reduce -> clip + Adam.  Capturable (no host values baked in)."""
        fp, st = self.flat, _stream()
        L.check(lib.uir_sumsq_partials(fp.grad.data_ptr(), self.n, self.slab.data_ptr(), _SUMSQ_BLOCKS, st), "uir_sumsq_partials")
        L.check(lib.uir_reduce_slab(self.slab.data_ptr(), _SUMSQ_BLOCKS, 1, self.sumsq.data_ptr(), 1, 0, 1.0, st), "uir_reduce_slab")
        a = self.args
        a.p, a.g, a.m, a.v = fp.flat.data_ptr(), fp.grad.data_ptr(), self.m.data_ptr(), self.v.data_ptr()
        a.sumsq, a.norm_out, a.n = self.sumsq.data_ptr(), self.norm.data_ptr(), self.n
        a.hyper = self.hyper.data_ptr()
        a.beta1, a.beta2, a.eps, a.max_norm = self.betas[0], self.betas[1], self.eps, self.max_norm
        L.check(lib.uir_clip_adam_step(C.byref(a), st), "uir_clip_adam_step")
        return self.norm

    def step(self, lr: float | None = None, grad_scale: float = 1.0) -> torch.Tensor:
        """Returns the (pre-clip) total gradient norm as a device tensor [1]; no host sync."""
        self.advance(lr, grad_scale)
        return self.launch()

    def state_dict(self):
        return {"step": self.t, "exp_avg": self.m, "exp_avg_sq": self.v, "lr": self.lr}

    def load_state_dict(self, sd):
        self.t = int(sd["step"])
        self.m.copy_(sd["exp_avg"])
        self.v.copy_(sd["exp_avg_sq"])
        self.lr = float(sd.get("lr", self.lr))
