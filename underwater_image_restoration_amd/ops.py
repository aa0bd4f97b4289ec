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


class _FflPlan:
    """FocalFrequencyLoss(loss_weight=1, alpha=1) of the reference (src/Losses/losses.py:45-46; focal_frequency_loss v0.3.0 with its
    defaults patch_factor 1, ave_spectrum / log_matrix / batch_matrix False) for one (B, H, W):
        D = fft2(pred - truth, norm="ortho");  w = |D| / max_{H,W}|D| per (image, channel), detached, NaN -> 0, clamped to [0, 1]
        loss = mean(w |D|^2)
    The package is not in the image and not vendored by the reference: this restates its published algorithm (PARITY UNPINNED).
    The two DFT axes run on the plan's own HWC4 buffers (uir_dft_axis*; |D| does not depend on the transform's sign convention, and
    the gradient of a real input is Re DFT(w conj(D)) with the SAME sign), everything else is five small kernels of csrc/loss.hip.
    The backward recomputes the spectrum: nothing is assumed about which tensors the forward last saw."""

    _cache: dict = {}

    @classmethod
    def get(cls, B, H, W, device):
        key = (B, H, W, str(device))
        if key not in cls._cache:
            if len(cls._cache) >= 4:
                cls._cache.pop(next(iter(cls._cache)))
            cls._cache[key] = cls(B, H, W, device)
        return cls._cache[key]

    def __init__(self, B, H, W, device):
        from .engine import Recorder
        from .model.block import record_axis_dft
        self.B, self.H, self.W = B, H, W
        r = self.rec = Recorder(device, scratch_floats=1 << 16)
        M, s = B * H * W, 1.0 / float(H * W) ** 0.5
        self.d4 = r.alloc(M, 4)
        ar, ai = record_axis_dft(r, "ffl.W", self.d4, None, B * H, W, 1, 4, 1.0, real_out=False)
        self.Dr, self.Di = record_axis_dft(r, "ffl.H", ar, ai, B, H, W, 4, s, real_out=False)
        self.nblk = max(1, min(64, (H * W + 4095) // 4096))
        self.mslab, self.mx2, self.sslab = r.alloc(B * self.nblk * 4), r.alloc(B * 4), r.alloc(B * self.nblk)
        self.Gr, self.Gi = r.alloc(M, 4), r.alloc(M, 4)
        r.cur = r.bwd
        hr, hi = record_axis_dft(r, "ffl.bwd.H", self.Gr, self.Gi, B, H, W, 4, 1.0, real_out=False)
        self.dd4, _ = record_axis_dft(r, "ffl.bwd.W", hr, hi, B * H, W, 1, 4, s, real_out=True)
        r.cur = r.fwd

    def _spectrum(self, pred, truth, st):
        B, HW = self.B, self.H * self.W
        L.check(lib.uir_ffl_diff(pred.data_ptr(), truth.data_ptr(), self.d4.data_ptr(), B, HW, st), "uir_ffl_diff")
        self.rec.run(self.rec.fwd)
        L.check(lib.uir_ffl_max(self.Dr.data_ptr(), self.Di.data_ptr(), self.mx2.data_ptr(), self.mslab.data_ptr(), self.nblk, B, HW, st), "uir_ffl_max")

    def forward(self, pred, truth, out_ptr: int):
        """writes the FFL mean to the device float at out_ptr"""
        st = _stream()
        B, HW = self.B, self.H * self.W
        self._spectrum(pred, truth, st)
        L.check(lib.uir_ffl_partials(self.Dr.data_ptr(), self.Di.data_ptr(), self.mx2.data_ptr(), self.sslab.data_ptr(), self.nblk, B, HW, st),
                "uir_ffl_partials")
        L.check(lib.uir_reduce_slab(self.sslab.data_ptr(), B * self.nblk, 1, out_ptr, 1, 0, 1.0 / (B * 3 * HW), st), "uir_reduce_slab")

    def backward(self, pred, truth, gscale_ptr: int, w_ptr, dpred, accumulate: bool, scale: float = 1.0):
        """dpred (+)= scale * gscale * (w or 1) * dFFL/dpred"""
        st = _stream()
        B, HW = self.B, self.H * self.W
        self._spectrum(pred, truth, st)
        coef = scale * 2.0 / (B * 3 * HW)         # d mean(w |D|^2) / dD = 2 w D / count (w detached); the DFT scale sits in the plan
        L.check(lib.uir_ffl_spectrum_grad(self.Dr.data_ptr(), self.Di.data_ptr(), self.mx2.data_ptr(), gscale_ptr, w_ptr, coef,
                                          self.Gr.data_ptr(), self.Gi.data_ptr(), B, HW, st), "uir_ffl_spectrum_grad")
        self.rec.run(self.rec.bwd)
        L.check(lib.uir_ffl_add_grad(self.dd4.data_ptr(), dpred.data_ptr(), B, HW, 1 if accumulate else 0, st), "uir_ffl_add_grad")


class _MsSsimPlan:
    """MS_SSIM(win_size=11, win_sigma=1.5, data_range=1, size_average=True, channel=3) of the reference (src/Losses/losses.py:43-44) as a
    differentiable loss term for one (B, H, W); pytorch_msssim is not in the image and not vendored: its published algorithm is restated
    (PARITY UNPINNED, checked against oracle/msssim_ref.py).  Five levels, weights (0.0448, 0.2856, 0.3001, 0.2363, 0.1333), K = (0.01,
    0.03); needs min(H, W) > 160 like the package.  The backward recomputes the forward (pooled images, level means)."""

    WEIGHTS = (0.0448, 0.2856, 0.3001, 0.2363, 0.1333)
    _cache: dict = {}

    @classmethod
    def get(cls, B, H, W, device):
        key = (B, H, W, str(device))
        if key not in cls._cache:
            if len(cls._cache) >= 4:
                cls._cache.pop(next(iter(cls._cache)))
            cls._cache[key] = cls(B, H, W, device)
        return cls._cache[key]

    def __init__(self, B, H, W, device):
        if min(H, W) <= (11 - 1) * 2 ** 4:
            raise ValueError(f"MS-SSIM: image sides must exceed {(11 - 1) * 2 ** 4} (4 downsamplings of an 11-tap window), got {H} x {W}")
        self.B, self.planes = B, B * 3
        f32 = dict(dtype=torch.float32, device=device)
        coords = torch.arange(11, dtype=torch.float32) - 5            # _fspecial_gauss_1d(11, 1.5), fp32 as in the package
        g = torch.exp(-(coords ** 2) / (2 * 1.5 ** 2))
        self.win = [float(v) for v in (g / g.sum())]
        self.sizes = [(H, W)]
        for _ in range(4):
            h, w = self.sizes[-1]
            ph, pw = h % 2, w % 2
            self.sizes.append(((h + 2 * ph - 2) // 2 + 1, (w + 2 * pw - 2) // 2 + 1))
        P = self.planes
        self.X = [None] + [torch.empty(P * h * w, **f32) for h, w in self.sizes[1:]]        # level 0 = the caller's tensors
        self.Y = [None] + [torch.empty(P * h * w, **f32) for h, w in self.sizes[1:]]
        self.dX = [torch.empty(P * h * w, **f32) for h, w in self.sizes]
        self.nblk = [max(1, min(64, ((h - 10) * (w - 10) + 4095) // 4096)) for h, w in self.sizes]
        self.slab = torch.empty(P * 64 * 2, **f32)
        self.means, self.dmeans, self.ms = torch.empty(5 * P * 2, **f32), torch.empty(5 * P * 2, **f32), torch.empty(1, **f32)
        n0 = P * (H - 10) * (W - 10)
        self.A, self.Bm, self.Cm = torch.empty(n0, **f32), torch.empty(n0, **f32), torch.empty(n0, **f32)
        self.wts = L.MsSsimWeights()
        for i, v in enumerate(self.WEIGHTS):
            self.wts.w[i] = v

    def _args(self, lvl, pred, truth):
        a = L.MsSsimArgs()
        a.X = pred.data_ptr() if lvl == 0 else self.X[lvl].data_ptr()
        a.Y = truth.data_ptr() if lvl == 0 else self.Y[lvl].data_ptr()
        a.slab = self.slab.data_ptr()
        a.planes, (a.H, a.W), a.nblk = self.planes, self.sizes[lvl], self.nblk[lvl]
        a.C1, a.C2 = 0.01 ** 2, 0.03 ** 2
        for i, v in enumerate(self.win):
            a.win[i] = v
        return a

    def _forward(self, pred, truth, st, ms_ptr=None):
        P = self.planes
        for lvl in range(5):
            a = self._args(lvl, pred, truth)
            L.check(lib.uir_msssim_level(C.byref(a), self.means.data_ptr() + 4 * (lvl * P * 2), st), "uir_msssim_level")
            if lvl < 4:
                h, w = self.sizes[lvl]
                L.check(lib.uir_avgpool2(a.X, self.X[lvl + 1].data_ptr(), P, h, w, st), "uir_avgpool2")
                L.check(lib.uir_avgpool2(a.Y, self.Y[lvl + 1].data_ptr(), P, h, w, st), "uir_avgpool2")
        L.check(lib.uir_msssim_combine(self.means.data_ptr(), self.dmeans.data_ptr(), ms_ptr or self.ms.data_ptr(), 5, P, C.byref(self.wts), st),
                "uir_msssim_combine")

    def forward(self, pred, truth, out_ptr: int):
        """writes ms_ssim(pred, truth) to the device float at out_ptr"""
        self._forward(pred, truth, _stream(), out_ptr)

    def backward(self, pred, truth, gscale_ptr: int, coef: float, dpred):
        """dpred += coef * gscale * d ms_ssim / d pred"""
        st = _stream()
        P = self.planes
        self._forward(pred, truth, st)
        for lvl in range(4, -1, -1):
            a = self._args(lvl, pred, truth)
            L.check(lib.uir_msssim_level_bwd(C.byref(a), self.dmeans.data_ptr() + 4 * (lvl * P * 2), self.A.data_ptr(), self.Bm.data_ptr(),
                                             self.Cm.data_ptr(), self.dX[lvl].data_ptr(), 0, st), "uir_msssim_level_bwd")
            if lvl < 4:
                h, w = self.sizes[lvl]
                L.check(lib.uir_avgpool2_bwd(self.dX[lvl + 1].data_ptr(), self.dX[lvl].data_ptr(), P, h, w, 1, st), "uir_avgpool2_bwd")
        L.check(lib.uir_axpy_dev(self.dX[0].data_ptr(), gscale_ptr, coef, dpred.data_ptr(), self.dX[0].numel(), st), "uir_axpy_dev")


def loss_sums(pred: torch.Tensor, truth: torch.Tensor, kind: int) -> torch.Tensor:
    """-> device tensor [4] = (loss(kind), PSNR, and the two detached weights of AdaptiveLuminanceCharbonnier - pass it to
    loss_grad as `aux`)."""
    pred, truth = _req(pred, "loss"), _req(truth, "loss")
    if pred.shape != truth.shape or pred.dim() != 4 or pred.shape[1] != 3:
        raise ValueError(f"loss expects matching [B,3,H,W] tensors, got {tuple(pred.shape)} / {tuple(truth.shape)}")
    if kind >= L.LOSS_MIX_FIRST:         # the mixes around the VGG16 perceptual term: composed from the single-term paths
        from .Losses import perceptual
        return perceptual.mix_sums(pred, truth, kind)
    B, _, H, W = pred.shape
    slab = torch.empty(_LOSS_BLOCKS * 8, dtype=torch.float32, device=pred.device)
    sums = torch.empty(8, dtype=torch.float32, device=pred.device)
    out = torch.empty(4, dtype=torch.float32, device=pred.device)
    st = _stream()
    L.check(lib.uir_loss_partials(pred.data_ptr(), truth.data_ptr(), slab.data_ptr(), _LOSS_BLOCKS, B, H * W, st), "uir_loss_partials")
    L.check(lib.uir_reduce_slab(slab.data_ptr(), _LOSS_BLOCKS, 8, sums.data_ptr(), 8, 0, 1.0, st), "uir_reduce_slab")
    if kind == L.LOSS_GRADIENT:          # Gradient_Loss: its own stencil pass; the mean lands in sums[5]
        if H < 3 or W < 3:
            raise ValueError("the gradient loss needs H, W >= 3 (valid 3x3 Laplacian)")
        gslab = torch.empty(_LOSS_BLOCKS, dtype=torch.float32, device=pred.device)
        L.check(lib.uir_gradloss_partials(pred.data_ptr(), truth.data_ptr(), gslab.data_ptr(), _LOSS_BLOCKS, B, H, W, st), "uir_gradloss_partials")
        L.check(lib.uir_reduce_slab(gslab.data_ptr(), _LOSS_BLOCKS, 1, sums.data_ptr() + 20, 1, 0, 1.0 / (B * 3 * (H - 2) * (W - 2)), st),
                "uir_reduce_slab")
    if kind in L.LOSS_FFL_KINDS:         # FocalFrequencyLoss mean -> sums[6]
        _FflPlan.get(B, H, W, pred.device).forward(pred, truth, sums.data_ptr() + 24)
    if kind in L.LOSS_MSSSIM_KINDS:      # MS-SSIM mean -> sums[7]
        _MsSsimPlan.get(B, H, W, pred.device).forward(pred, truth, sums.data_ptr() + 28)
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


def loss_grad(pred: torch.Tensor, truth: torch.Tensor, gscale: torch.Tensor, kind: int, aux: "torch.Tensor | None" = None) -> torch.Tensor:
    """dL/dpred * gscale.  aux = the tensor loss_sums returned for the same pair (needed by AdaptiveLuminanceCharbonnier: its weights
    are functions of the two loss values, detached - src/Losses/losses.py:135-138)."""
    pred, truth = _req(pred, "loss"), _req(truth, "loss")
    if kind >= L.LOSS_MIX_FIRST:
        from .Losses import perceptual
        return perceptual.mix_grad(pred, truth, gscale, kind, aux)
    B, _, H, W = pred.shape
    gs = _req(gscale.reshape(1), "loss")
    dpred = torch.empty_like(pred)
    if kind == L.LOSS_GRADIENT:
        L.check(lib.uir_gradloss_bwd(pred.data_ptr(), truth.data_ptr(), gs.data_ptr(), dpred.data_ptr(), B, H, W, _stream()), "uir_gradloss_bwd")
        return dpred
    if kind in L.LOSS_ADAPTIVE_KINDS and aux is None:
        aux = loss_sums(pred, truth, kind)
    L.check(lib.uir_loss_bwd(pred.data_ptr(), truth.data_ptr(), gs.data_ptr(), aux.data_ptr() if aux is not None else None, dpred.data_ptr(),
                             kind, B, H * W, _stream()), "uir_loss_bwd")
    if kind in L.LOSS_FFL_KINDS:         # + the FocalFrequencyLoss term (weighted by aux[2] in the adaptive pair, 0.025 in ssimFFLCharbonnier)
        wp = aux.data_ptr() + 8 if kind in L.LOSS_ADAPTIVE_KINDS else None
        _FflPlan.get(B, H, W, pred.device).backward(pred, truth, gs.data_ptr(), wp, dpred, accumulate=True,
                                                    scale=0.025 if kind == L.LOSS_KINDS["ssimFFLCharbonnier"] else 1.0)
    if kind in L.LOSS_MSSSIM_KINDS:      # + the (1 - ms_ssim) term
        _MsSsimPlan.get(B, H, W, pred.device).backward(pred, truth, gs.data_ptr(), -1.0, dpred)
    return dpred


class FusedClipAdam:
    """clip_grad_norm_(max_norm) + Adam(lr, betas=(0.9,0.999), eps=1e-8) - the step tail of
    src/ModelTrainer.py:87-88,199 - as two kernels over the flat live-parameter buffer.
    ``weight_decay > 0`` selects torch.optim.AdamW's decoupled decay (:201, default 0.01 there).

    Per-step scalars {lr, bc1, bc2, gscale} travel BY VALUE in the launch arguments (copied at launch time), so a host
    that runs many steps ahead of the GPU can never change a queued step.  Only hipGraph replay reads them from device
    memory, where ``uir_adam_set_hyper`` (again by-value arguments, stream ordered) stores them before each replay."""

    def __init__(self, flat, lr: float = 3e-4, betas=(0.9, 0.999), eps: float = 1e-8, max_norm: float = 1.0,
                 weight_decay: float = 0.0):
        self.flat = flat
        self.n = flat.n_live
        self.lr, self.betas, self.eps, self.max_norm, self.weight_decay = lr, betas, eps, max_norm, weight_decay
        self.initial_lr = lr
        dev = flat.device
        self.m = torch.zeros(self.n, dtype=torch.float32, device=dev)
        self.v = torch.zeros(self.n, dtype=torch.float32, device=dev)
        self.slab = torch.empty(_SUMSQ_BLOCKS, dtype=torch.float32, device=dev)
        self.sumsq = torch.zeros(1, dtype=torch.float32, device=dev)
        self.norm = torch.zeros(1, dtype=torch.float32, device=dev)
        self.t = 0
        self.args = L.AdamArgs()
        self.hyper = torch.zeros(4, dtype=torch.float32, device=dev)      # hipGraph replay only

    def _scalars(self, lr, grad_scale):
        if lr is not None:
            self.lr = lr
        self.t += 1
        return self.lr, 1.0 - self.betas[0] ** self.t, 1.0 - self.betas[1] ** self.t, grad_scale

    def advance(self, lr: float | None = None, grad_scale: float = 1.0) -> None:
        """hipGraph path: bump the step count and store this step's scalars in `hyper` with a by-value launch on the
        current stream (ordered before the replay that follows; no host buffer the host could overwrite early)."""
        lr_, bc1, bc2, gs = self._scalars(lr, grad_scale)
        L.check(lib.uir_adam_set_hyper(self.hyper.data_ptr(), lr_, bc1, bc2, gs, _stream()), "uir_adam_set_hyper")

    def launch(self, scalars=None, grad: "torch.Tensor | None" = None) -> torch.Tensor:
        """Device side: sum of squares -> reduce -> clip + Adam.  scalars=None: read them from `hyper` (capturable).
        grad: the flat gradient to consume (default: the plan's bucket; the micro-batch accumulator otherwise)."""
        fp, st = self.flat, _stream()
        grad = fp.grad if grad is None else grad
        L.check(lib.uir_sumsq_partials(grad.data_ptr(), self.n, self.slab.data_ptr(), _SUMSQ_BLOCKS, st), "uir_sumsq_partials")
        L.check(lib.uir_reduce_slab(self.slab.data_ptr(), _SUMSQ_BLOCKS, 1, self.sumsq.data_ptr(), 1, 0, 1.0, st), "uir_reduce_slab")
        a = self.args
        a.p, a.g, a.m, a.v = fp.flat.data_ptr(), grad.data_ptr(), self.m.data_ptr(), self.v.data_ptr()
        a.sumsq, a.norm_out, a.n = self.sumsq.data_ptr(), self.norm.data_ptr(), self.n
        a.beta1, a.beta2, a.eps, a.max_norm = self.betas[0], self.betas[1], self.eps, self.max_norm
        a.weight_decay = self.weight_decay
        if scalars is None:
            a.hyper = self.hyper.data_ptr()
            a.lr, a.bc1, a.bc2, a.gscale = self.lr, 1.0, 1.0, 1.0
        else:
            a.hyper = None
            a.lr, a.bc1, a.bc2, a.gscale = scalars
        L.check(lib.uir_clip_adam_step(C.byref(a), st), "uir_clip_adam_step")
        return self.norm

    def step(self, lr: float | None = None, grad_scale: float = 1.0, grad: "torch.Tensor | None" = None) -> torch.Tensor:
        """Returns the (pre-clip) total gradient norm as a device tensor [1]; no host sync, no host staging."""
        return self.launch(self._scalars(lr, grad_scale), grad)

    # ---- checkpoint: the layout torch.optim.Adam.state_dict() writes (src/ModelTrainer.py:176-181 saves exactly that) ----
    def state_dict(self):
        """{'state': {i: {'step', 'exp_avg', 'exp_avg_sq'}}, 'param_groups': [...]} with i = index in model.parameters()
        order; the 63 parameters the reference never uses have no state entry (torch creates state lazily, on first grad)."""
        fp = self.flat
        state = {}
        for i, name in enumerate(fp.shapes):
            o, n = fp.slots[name]
            if o + n > self.n or self.t == 0:
                continue
            state[i] = {"step": torch.tensor(float(self.t)),
                        "exp_avg": self.m[o:o + n].view(fp.shapes[name]).clone(),
                        "exp_avg_sq": self.v[o:o + n].view(fp.shapes[name]).clone()}
        group = {"lr": self.lr, "betas": tuple(self.betas), "eps": self.eps, "weight_decay": self.weight_decay,
                 "amsgrad": False, "maximize": False, "foreach": None, "capturable": False, "differentiable": False,
                 "fused": None, "decoupled_weight_decay": self.weight_decay > 0, "initial_lr": self.initial_lr,
                 "params": list(range(len(fp.shapes)))}
        return {"state": state, "param_groups": [group]}

    def load_state_dict(self, sd):
        fp = self.flat
        if "param_groups" not in sd:
            raise ValueError("expected a torch.optim.Adam-style state_dict ({'state', 'param_groups'})")
        g = sd["param_groups"][0]
        self.lr = float(g["lr"])
        self.initial_lr = float(g.get("initial_lr", self.initial_lr))
        self.betas, self.eps = tuple(g.get("betas", self.betas)), float(g.get("eps", self.eps))
        self.weight_decay = float(g.get("weight_decay", self.weight_decay))
        names = list(fp.shapes)
        steps = set()
        self.m.zero_()
        self.v.zero_()
        for i, st in sd["state"].items():
            name = names[int(i)]
            o, n = fp.slots[name]
            if o + n > self.n:
                raise ValueError(f"optimizer state for '{name}', which is not on the live path")
            self.m[o:o + n].copy_(st["exp_avg"].reshape(-1))
            self.v[o:o + n].copy_(st["exp_avg_sq"].reshape(-1))
            steps.add(int(float(st["step"])))
        if len(steps) > 1:
            raise ValueError(f"per-parameter step counts differ ({sorted(steps)}): not a state this optimizer can resume")
        self.t = steps.pop() if steps else 0


def flat_accumulate(src: torch.Tensor, dst: torch.Tensor, n: int, accumulate: bool) -> None:
    """dst[:n] (+)= src[:n] over flat fp32 buffers (n a multiple of 4), through uir_copy_cols on rows of 4 floats."""
    assert n % 4 == 0 and src.numel() >= n and dst.numel() >= n
    L.check(lib.uir_copy_cols(src.data_ptr(), 4, 0, dst.data_ptr(), 4, 0, n // 4, 4, 1 if accumulate else 0, _stream()), "uir_copy_cols")


def axpy_scalar(dst: torch.Tensor, src: torch.Tensor, w: float, first: bool) -> None:
    """dst[0] = (0 if first else dst[0]) + w * src[0] on the device (loss bookkeeping of micro-batched steps)."""
    if first:
        torch.mul(src.reshape(1), w, out=dst)
    else:
        dst.add_(src.reshape(1), alpha=w)


class MultiFlatClipAdam:
    """clip_grad_norm_(max_norm) + Adam / AdamW over SEVERAL flat parameter buffers (one per launch-plan module of a composed
    network: NewBigFRFNModel has 27) with ONE global gradient norm - the same step tail as FusedClipAdam
    (src/ModelTrainer.py:87-88,197-204): per buffer a sum-of-squares pass into one slab, one reduction, per buffer the fused
    clip + Adam kernel with that shared sum.  Per-step scalars by value."""

    def __init__(self, flats, lr: float = 3e-4, betas=(0.9, 0.999), eps: float = 1e-8, max_norm: float = 1.0, weight_decay: float = 0.0):
        self.flats = list(flats)
        self.lr, self.initial_lr, self.betas, self.eps, self.max_norm, self.weight_decay = lr, lr, betas, eps, max_norm, weight_decay
        dev = self.flats[0].device
        self.m = [torch.zeros(f.total, dtype=torch.float32, device=dev) for f in self.flats]
        self.v = [torch.zeros(f.total, dtype=torch.float32, device=dev) for f in self.flats]
        self.slab = torch.empty(len(self.flats) * _SUMSQ_BLOCKS, dtype=torch.float32, device=dev)
        self.sumsq = torch.zeros(1, dtype=torch.float32, device=dev)
        self.norm = torch.zeros(1, dtype=torch.float32, device=dev)
        self.t = 0
        self._args = [L.AdamArgs() for _ in self.flats]

    def step(self, lr: "float | None" = None, grad_scale: float = 1.0) -> torch.Tensor:
        if lr is not None:
            self.lr = lr
        self.t += 1
        bc1, bc2 = 1.0 - self.betas[0] ** self.t, 1.0 - self.betas[1] ** self.t
        st = _stream()
        for i, fp in enumerate(self.flats):
            # live parameters only: torch.optim skips `p.grad is None` (no moments, no weight decay for the parameters the reference
            # never uses - 88 tensors of NewBigFRFNModel); FlatParams keeps them behind n_live
            L.check(lib.uir_sumsq_partials(fp.grad.data_ptr(), fp.n_live, self.slab.data_ptr() + 4 * i * _SUMSQ_BLOCKS, _SUMSQ_BLOCKS, st),
                    "uir_sumsq_partials")
        L.check(lib.uir_reduce_slab(self.slab.data_ptr(), len(self.flats) * _SUMSQ_BLOCKS, 1, self.sumsq.data_ptr(), 1, 0, 1.0, st),
                "uir_reduce_slab")
        for a, fp, m, v in zip(self._args, self.flats, self.m, self.v):
            a.p, a.g, a.m, a.v = fp.flat.data_ptr(), fp.grad.data_ptr(), m.data_ptr(), v.data_ptr()
            a.sumsq, a.norm_out, a.n, a.hyper = self.sumsq.data_ptr(), self.norm.data_ptr(), fp.n_live, None
            a.beta1, a.beta2, a.eps, a.max_norm, a.weight_decay = self.betas[0], self.betas[1], self.eps, self.max_norm, self.weight_decay
            a.lr, a.bc1, a.bc2, a.gscale = self.lr, bc1, bc2, grad_scale
            L.check(lib.uir_clip_adam_step(C.byref(a), st), "uir_clip_adam_step")
        return self.norm

    def state_dict(self):
        """torch.optim.Adam layout over the concatenation of the modules' parameters in buffer order (= model.parameters() order
        when the buffers are given in registration order); the parameters that never receive a gradient have no entry, as in
        torch (state is created on the first gradient)."""
        state, params, i = {}, [], 0
        for fp, m, v in zip(self.flats, self.m, self.v):
            for name in fp.shapes:
                o, n = fp.slots[name]
                if self.t > 0 and name not in fp.dead_names:
                    state[i] = {"step": torch.tensor(float(self.t)), "exp_avg": m[o:o + n].view(fp.shapes[name]).clone(),
                                "exp_avg_sq": v[o:o + n].view(fp.shapes[name]).clone()}
                params.append(i)
                i += 1
        group = {"lr": self.lr, "betas": tuple(self.betas), "eps": self.eps, "weight_decay": self.weight_decay, "amsgrad": False,
                 "maximize": False, "foreach": None, "capturable": False, "differentiable": False, "fused": None,
                 "decoupled_weight_decay": self.weight_decay > 0, "initial_lr": self.initial_lr, "params": params}
        return {"state": state, "param_groups": [group]}

    def load_state_dict(self, sd):
        """Resume from a torch.optim.Adam / AdamW-style state_dict over model.parameters() order (what SaveModel writes)."""
        if "param_groups" not in sd:
            raise ValueError("expected a torch.optim.Adam-style state_dict ({'state', 'param_groups'})")
        g = sd["param_groups"][0]
        self.lr = float(g["lr"])
        self.initial_lr = float(g.get("initial_lr", self.initial_lr))
        self.betas, self.eps = tuple(g.get("betas", self.betas)), float(g.get("eps", self.eps))
        self.weight_decay = float(g.get("weight_decay", self.weight_decay))
        index = []                                            # parameter index -> (buffer, name)
        for k, fp in enumerate(self.flats):
            index += [(k, name) for name in fp.shapes]
        steps = set()
        for m, v in zip(self.m, self.v):
            m.zero_()
            v.zero_()
        for i, st in sd["state"].items():
            k, name = index[int(i)]
            fp = self.flats[k]
            if name in fp.dead_names:
                raise ValueError(f"optimizer state for '{name}', which never receives a gradient")
            o, n = fp.slots[name]
            self.m[k][o:o + n].copy_(st["exp_avg"].reshape(-1))
            self.v[k][o:o + n].copy_(st["exp_avg_sq"].reshape(-1))
            steps.add(int(float(st["step"])))
        if len(steps) > 1:
            raise ValueError(f"per-parameter step counts differ ({sorted(steps)}): not a state this optimizer can resume")
        self.t = steps.pop() if steps else 0
