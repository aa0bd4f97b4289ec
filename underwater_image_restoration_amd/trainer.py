"""The fused train step of the hot path (src/ModelTrainer.py:78-90): zero_grad -> forward ->
loss -> backward -> clip_grad_norm_(1.0) -> Adam.step, as launch lists + three reductions on
one HIP stream with no host synchronisation.  Under data parallelism (one process per GPU) the
flat gradient buffer is all-reduced once over RCCL/xGMI between backward and the optimizer.
"""
from __future__ import annotations

import torch
import torch.distributed as dist

from . import _lib as L
from . import knobs
from . import ops
from . import parallel


def debug_check_streams(device) -> None:
    """UIR_DEBUG=1: the part of INTEGRATION.md section 6 a process CAN check.  The step's kernels must not share the chip with torch's own
    kernels (gfx950 packed-f32 op_sel erratum); stream order guarantees that for everything on the stream the step is called
    on.  The usual way to break it is to call the step under `torch.cuda.stream(s)` while the device's default stream still has torch work in
    flight: that is visible (`default_stream().query()`), and refused here.  Streams created by other libraries stay invisible."""
    if not knobs.flag("UIR_DEBUG") or torch.device(device).type != "cuda":
        return
    cur, dflt = torch.cuda.current_stream(device), torch.cuda.default_stream(device)
    if cur != dflt and not dflt.query():
        raise RuntimeError("UIR_DEBUG: the train step was launched on a non-default stream while the device's default stream has work in flight - "
                           "torch kernels may then run beside the step's bf16-MFMA kernels (INTEGRATION.md section 6): order the streams "
                           "(cur.wait_stream(default)) or run the step on the default stream")


class TrainStep:
    def __init__(self, model, loss_name: str = "L1", lr: float = 3e-4, max_norm: float = 1.0, process_group=None,
                 weight_decay: float = 0.0, micro_batch: "int | None" = None):
        if loss_name not in L.LOSS_KINDS:
            raise ValueError(f"Unsupported loss: {loss_name}")
        self.model = model
        self.kind = L.LOSS_KINDS[loss_name]
        self.flat = model.flat_params()
        if self.flat.device.type == "cuda" and self.flat.device.index not in (None, torch.cuda.current_device()):
            # every launch goes to torch.cuda.current_stream(): it must belong to the device that owns the buffers
            raise RuntimeError(f"model lives on {self.flat.device} but the current device is cuda:{torch.cuda.current_device()}: "
                               "call torch.cuda.set_device(...) first (one process per GPU)")
        self.opt = ops.FusedClipAdam(self.flat, lr=lr, max_norm=max_norm, weight_decay=weight_decay)
        self.pg = process_group
        self.micro_batch = micro_batch          # None: the largest divisor of the batch whose plan fits in HBM
        self._acc = None                        # flat gradient accumulator of micro-batched steps
        self.world = parallel.world_size(process_group)
        if self.world > 1:
            parallel.broadcast_flat_params(self.flat.flat, process_group)
        self._one = torch.ones(1, dtype=torch.float32, device=self.flat.device)
        if knobs.get("UIR_MIXED") == "fp16":
            raise NotImplementedError("UIR_MIXED=fp16 needs the loss scaling of the module networks' train step (AST / NewBigFRFNModel); "
                                      "the SpectralTransformer step has none - use bf16, or unset it (its f32 path is the measured one)")
        self._graph_on = knobs.flag("UIR_GRAPH") and self.flat.device.type == "cuda"
        # opt-in (UIR_GRAPH=1): on ROCm 7.2 replaying the ~1400-node graph measured 62.2 ms/step vs 57.9 eager
        # a captured graph is stored ON its plan (plan._graph): it can never outlive the buffers it was recorded over

    def __call__(self, raw: torch.Tensor, ref: torch.Tensor, lr: float | None = None):
        """One optimisation step on a device-resident batch.  Returns (loss, grad_norm) as
        device tensors (reading them is the only synchronisation, and it is the caller's)."""
        debug_check_streams(self.flat.device)
        return self._step(raw, ref, lr)

    # ---- memory: one plan keeps every activation of its batch (62 GB at 16 x 256 x 256); the reference's own default run
    # (16 x 1024 x 1024, DataLoader.py:14) is ~1 TB that way, so large batches run as micro-batches of the SAME plan with
    # gradient accumulation.  Exact: the network has no cross-sample statistics (no BatchNorm; attention is per sample).
    def micro_batch_for(self, B: int, H: int, W: int) -> int:
        if self.micro_batch is not None:
            mb = int(self.micro_batch)
            if mb < 1 or B % mb:
                raise ValueError(f"micro_batch={mb} must divide the batch size {B}")
            return mb
        if self.flat.device.type != "cuda":
            return B
        from .engine import plan_bytes_estimate, device_bytes_available, plan_fits_addressing
        shape_cached = any(k[:3] == (B, H, W) and k[3] for k in self.model._plans)
        if shape_cached:
            return B
        have = device_bytes_available(self.flat.device) + sum(p.bytes for p in self.model._plans.values())
        for mb in sorted((d for d in range(1, B + 1) if B % d == 0), reverse=True):
            if plan_fits_addressing(mb, H, W) and plan_bytes_estimate(mb, H, W) <= 0.95 * have:
                return mb
        raise RuntimeError(f"SpectralTransformer(HIP): even one {H}x{W} image needs ~{plan_bytes_estimate(1, H, W) / 2**30:.1f} GiB "
                           f"of HBM for its train plan ({have / 2**30:.1f} GiB available)")

    def check_memory(self, B: int, H: int, W: int) -> int:
        """Fail before the first batch, with the numbers, if the configuration cannot run; returns the micro-batch."""
        mb = self.micro_batch_for(B, H, W)
        if mb != B and parallel.rank() == 0:
            print(f"[uir] batch {B} at {H}x{W} runs as {B // mb} micro-batches of {mb} (gradient accumulation; one plan keeps "
                  f"every activation of its micro-batch in HBM)")
        return mb

    def _weights(self, mb: int, B: int) -> float:
        """Weight of one micro-batch's loss in the full-batch loss: b/B for plain means; (b/B)^2 for the losses the
        reference also divides by the batch size (L1, L2, L1withColor / L1ColorLum: src/Losses/losses.py:57,65,74,78)."""
        if self.kind in L.LOSS_ADAPTIVE_KINDS and mb != B:
            raise NotImplementedError("the Adaptive* losses weight their two terms by their values over the WHOLE batch "
                                      "(losses.py:135-138): it cannot be accumulated over micro-batches")
        f = mb / B
        return f * f if self.kind in (0, 1, 4) else f

    def _step_micro(self, raw, ref, lr, mb):
        B = raw.shape[0]
        n = self.flat.n_live
        if self._acc is None:
            self._acc = torch.zeros(self.flat.total, dtype=torch.float32, device=self.flat.device)
            self._wdev = torch.zeros(1, dtype=torch.float32, device=self.flat.device)
            self._loss_acc = torch.zeros(1, dtype=torch.float32, device=self.flat.device)
        w = self._weights(mb, B)
        self._wdev.fill_(w)
        for i, o in enumerate(range(0, B, mb)):
            r, f = raw[o:o + mb], ref[o:o + mb]
            plan = self.model._plan_for(r)
            out = plan.forward(r)
            res = ops.loss_sums(out, f, self.kind)
            dout = ops.loss_grad(out, f, self._wdev, self.kind, aux=res)
            plan.backward(dout)
            self.model._fwd_serial += 1
            ops.flat_accumulate(self.flat.grad, self._acc, n, accumulate=i > 0)
            ops.axpy_scalar(self._loss_acc, res[0], w, first=i == 0)
        scale = parallel.allreduce_flat_grads(self._acc, n, self.pg)
        norm = self.opt.step(lr=lr, grad_scale=scale, grad=self._acc)
        return self._loss_acc[0].clone(), norm[0]

    def _step(self, raw, ref, lr):
        mb = self.micro_batch_for(raw.shape[0], raw.shape[2], raw.shape[3])
        if mb != raw.shape[0]:
            return self._step_micro(raw, ref, lr, mb)
        plan = self.model._plan_for(raw)
        if self._graph_on and self.world == 1:
            g = getattr(plan, "_graph", None)
            if g is None:
                plan._eager_calls = getattr(plan, "_eager_calls", 0) + 1
                if plan._eager_calls > 2:               # plan, streams and allocator pools are warm: capture
                    g = self._capture(plan, ref)
            if g is not None:
                plan.in_nchw.copy_(raw)
                g["ref"].copy_(ref)
                self.opt.advance(lr=lr, grad_scale=1.0)
                g["graph"].replay()
                self.model._fwd_serial += 1
                return g["res"][0], self.opt.norm[0]
        out = plan.forward(raw)
        res = ops.loss_sums(out, ref, self.kind)
        dout = ops.loss_grad(out, ref, self._one, self.kind, aux=res)
        plan.backward(dout)
        self.model._fwd_serial += 1
        # single flat bucket of the live gradients (8.8 MB): one all-reduce over xGMI
        scale = parallel.allreduce_flat_grads(self.flat.grad, self.flat.n_live, self.pg)
        norm = self.opt.step(lr=lr, grad_scale=scale)
        return res[0], norm[0]

    def _capture(self, plan, ref):
        """Capture forward list + loss + backward list (+ their side/bulk stream forks and joins) + clip/Adam
        into ONE hipGraph.  Inputs are the plan's static image buffer and a static reference buffer; the
        per-step scalars come from `opt.hyper` in device memory."""
        try:
            ref_s = torch.empty_like(ref)
            ref_s.copy_(ref)
            graph = torch.cuda.CUDAGraph()
            torch.cuda.synchronize()
            with torch.cuda.graph(graph):
                plan.run(plan.fwd)
                res = ops.loss_sums(plan.out_nchw, ref_s, self.kind)
                dout = ops.loss_grad(plan.out_nchw, ref_s, self._one, self.kind, aux=res)
                plan.backward(dout)
                self.opt.launch()
            g = {"graph": graph, "ref": ref_s, "res": res}
        except Exception as e:   # keep training on the eager launch lists (same kernels, same results)
            import warnings
            warnings.warn(f"hipGraph capture failed ({type(e).__name__}: {e}); staying on eager launch lists")
            self._graph_on = False
            torch.cuda.synchronize()
            return None
        plan._graph = g
        return g

    @torch.no_grad()
    def evaluate(self, raw: torch.Tensor, ref: torch.Tensor, with_ssim: bool = False):
        """Validation batch (src/ModelTrainer.py:134-158): -> (loss, PSNR) device tensors, plus torchSSIM(ref, out)
        (:23-24,149) with ``with_ssim`` (needs H, W >= 11).  A batch whose forward plan does not fit in HBM is evaluated in
        micro-batches: the loss recombines as in training, PSNR through the mean squared error it is a function of
        (20 log10(1/rmse) over the whole batch, :17-21), SSIM is a mean over samples."""
        B = raw.shape[0]
        mb = self._eval_micro_batch(B, raw.shape[2], raw.shape[3])
        if mb == B:
            plan = self.model._plan_for(raw, want_backward=False)
            out = plan.forward(raw)
            self.model._fwd_serial += 1
            res = ops.loss_sums(out, ref, self.kind)
            if with_ssim:
                return res[0], res[1], ops.ssim(ref, out, data_range=1.0)
            return res[0], res[1]
        w = self._weights(mb, B)
        loss = mse = ss = None
        for o in range(0, B, mb):
            r, f = raw[o:o + mb], ref[o:o + mb]
            plan = self.model._plan_for(r, want_backward=False)
            out = plan.forward(r)
            self.model._fwd_serial += 1
            res = ops.loss_sums(out, f, self.kind)
            l_i, m_i = res[0] * w, torch.pow(10.0, -res[1] / 10.0) * (mb / B)
            loss = l_i if loss is None else loss + l_i
            mse = m_i if mse is None else mse + m_i
            if with_ssim:
                s_i = ops.ssim(f, out, data_range=1.0) * (mb / B)
                ss = s_i if ss is None else ss + s_i
        psnr = -10.0 * torch.log10(mse)
        return (loss, psnr, ss) if with_ssim else (loss, psnr)

    def _eval_micro_batch(self, B: int, H: int, W: int) -> int:
        if self.micro_batch is not None:
            return min(B, self.micro_batch) if B % min(B, self.micro_batch) == 0 else 1
        if any(k[:3] == (B, H, W) for k in self.model._plans):
            return B
        if self.flat.device.type != "cuda":
            return B
        from .engine import plan_bytes_estimate, device_bytes_available, plan_fits_addressing
        have = device_bytes_available(self.flat.device) + sum(p.bytes for p in self.model._plans.values())
        divisors = sorted((d for d in range(1, B + 1) if B % d == 0), reverse=True)
        for mb in divisors:                    # a plan of this batch's shape that already exists (e.g. the training plan) first
            if any(k[:3] == (mb, H, W) for k in self.model._plans):
                return mb
        for mb in divisors:
            if plan_fits_addressing(mb, H, W) and plan_bytes_estimate(mb, H, W, False) <= 0.95 * have:
                return mb
        raise RuntimeError(f"SpectralTransformer(HIP): one {H}x{W} image does not fit in HBM for evaluation")


class ModuleTrainStep:
    """The train step (src/ModelTrainer.py:78-90) for a network COMPOSED of launch-plan modules (``--arch NewBigFRFNModel``: 27 of
    them, each one autograd node): forward through autograd, loss and its gradient by the HIP loss kernels, ``backward`` runs every
    module's backward launch list and leaves the gradients in the modules' flat buffers - views of ONE bucket -, (one all-reduce of
    that bucket under data parallelism,) one global-norm clip + Adam over all buffers (ops.MultiFlatClipAdam).  No host synchronisation inside."""

    def __init__(self, model, loss_name: str = "L1", lr: float = 3e-4, max_norm: float = 1.0, process_group=None, weight_decay: float = 0.0,
                 micro_batch: "int | None" = None):
        if loss_name not in L.LOSS_KINDS:
            raise ValueError(f"Unsupported loss: {loss_name}")
        self.model, self.kind, self.pg = model, L.LOSS_KINDS[loss_name], process_group
        self.micro_batch = micro_batch          # None: the largest divisor of the batch whose plans fit in HBM
        self._acc = None
        self._mb_cache = {}
        dev = next(model.parameters()).device
        if dev.type != "cuda":
            raise RuntimeError("the MI355X path needs the GPU ('cuda' device in PyTorch-ROCm); there is no CPU fallback")
        self.mods = model.plan_modules()
        owned = sum(len(list(m.parameters())) for m in self.mods)
        if owned != len(list(model.parameters())):
            raise RuntimeError("every parameter of the network must belong to a launch-plan module")
        for m in self.mods:
            m._ensure_flat(dev)
            m._flat_grads_only = True
            # from here on a module's output IS its plan's output buffer (a second call overwrites the first result): the contract
            # of a network driven by this step object, and it stays set for the modules' lifetime - clone an output to keep it
            m._alias_io = True
        self.flats = [m._flat for m in self.mods]
        # ONE gradient bucket for the whole network: every module's flat gradient buffer becomes a view of it, so data parallelism
        # is one all-reduce per step (AST: 28 modules, 80 MB) instead of one per module.  Plans record raw gradient
        # pointers when they are built: none may exist yet with the old buffers.
        self.bucket = torch.zeros(sum(fp.total for fp in self.flats), dtype=torch.float32, device=dev)
        off = 0
        for m, fp in zip(self.mods, self.flats):
            fp.grad = self.bucket[off:off + fp.total]
            m._plans.clear()
            off += fp.total
        self.world = parallel.world_size(process_group)
        if self.world > 1:
            for fp in self.flats:
                parallel.broadcast_flat_params(fp.flat, process_group)
        self.opt = ops.MultiFlatClipAdam(self.flats, lr=lr, max_norm=max_norm, weight_decay=weight_decay)
        # UIR_MIXED=fp16 (knobs.py): the gradient operands of the wide contractions are rounded to fp16, whose normal range ends at
        # 6e-5 - a mean-type loss at 4 x 512 x 512 starts its backward at 3e-7.  Loss scaling as in torch.cuda.amp: the loss gradient
        # is multiplied by GradScaler's initial scale 2^16 (exact in f32; every backward kernel is linear in it) and the optimizer's
        # gradient factor divides it out before the norm and the clip.  The scale is STATIC: the fp16 conversion saturates at
        # +-65504 instead of producing infinities (csrc/tiled.hip lp_pack2), there is no skip-and-back-off.  bf16 needs no scale.
        self.loss_scale = 65536.0 if knobs.get("UIR_MIXED") == "fp16" else 1.0
        self._one = torch.full((1,), self.loss_scale, dtype=torch.float32, device=dev)

    # ---- memory: every module's plan keeps all its activations.  The per-pixel figure is a constant of the architecture
    # (model.PLAN_BYTES_PER_PIXEL_TRAIN / _EVAL, held to the real plan bytes by tests/test_frfn_net.py / test_ast.py): AST needs
    # ~19 GB per 512 x 512 image, ~67 GB per image at the reference's default 1024 x 1024 (DataLoader.py:14), so a batch that does
    # not fit runs as micro-batches with gradient accumulation - exact in eval mode (no cross-sample statistics); in training mode
    # the stochastic-depth factors are drawn per micro-batch (another, equally valid draw).
    def plan_bytes(self) -> int:
        return sum(p.bytes for m in self.mods for p in m._plans.values())

    def plan_bytes_estimate(self, B: int, H: int, W: int, train: bool = True) -> int:
        per = getattr(self.model, "PLAN_BYTES_PER_PIXEL_TRAIN" if train else "PLAN_BYTES_PER_PIXEL_EVAL", None)
        if per is None:
            raise RuntimeError(f"{type(self.model).__name__} does not declare its plan memory per pixel")
        fixed = getattr(self.model, "PLAN_BYTES_FIXED_TRAIN" if train else "PLAN_BYTES_FIXED_EVAL", 0)
        return int(per) * B * H * W + int(fixed)

    def micro_batch_for(self, B: int, H: int, W: int, train: bool = True) -> int:
        if self.micro_batch is not None:
            mb = int(self.micro_batch)
            if mb < 1 or B % mb:
                raise ValueError(f"micro_batch={mb} must divide the batch size {B}")
            return mb
        key = (B, H, W, train)
        if key in self._mb_cache:                 # decided once per shape (the plans exist from then on)
            return self._mb_cache[key]
        if not train:
            # validation: a training plan of the same micro-batch shape serves a forward too (TokenModule._plan_for), so prefer a
            # micro-batch the training step already runs - no second set of plans beside plans that fill the HBM
            for (b2, h2, w2, t2), mb2 in self._mb_cache.items():
                if t2 and (h2, w2) == (H, W) and B % mb2 == 0:
                    self._mb_cache[key] = mb2
                    return mb2
        from . import engine
        dev = self.bucket.device

        def pick(have):
            for mb in sorted((d for d in range(1, B + 1) if B % d == 0), reverse=True):
                if self.plan_bytes_estimate(mb, H, W, train) <= 0.92 * have:
                    return mb
            return None
        # Only memory that is FREE counts: the modules keep the plans of other shapes alive (TokenModule._plan_for evicts by count,
        # never for memory), so a new shape sized against "free + existing plans" ran out of memory while recording its plans
        # (e.g. a validation tail batch beside training plans at 92 % of the HBM).  If the existing plans are in the way - the new
        # shape would get a smaller micro-batch than an empty device allows, or none - they are dropped first (rebuilt on demand,
        # their cached decisions with them) and the shape is sized against what is then really free.
        #  A VALIDATION shape that fits beside the training plans with a smaller micro-batch takes that micro-batch: dropping the
        # training plans for it made every train -> val -> train transition rebuild all plans, buffers and twiddle tables, silently, each epoch.
        free = engine.device_bytes_available(dev)
        mb, best = pick(free), pick(free + self.plan_bytes())
        if best is not None and (mb is None or (mb < best and train)):
            held = self.plan_bytes()
            self.drop_plans()
            free = engine.device_bytes_available(dev)
            mb = pick(free)
            if parallel.rank() == 0:
                print(f"[uir] {type(self.model).__name__}: dropped the launch plans of other shapes ({held / 2**30:.1f} GiB) to fit {'training' if train else 'validation'} "
                      f"batch {B} at {H}x{W} (micro-batch {mb}); they are rebuilt on demand", flush=True)
        if mb is None:
            raise RuntimeError(f"{type(self.model).__name__}(HIP): even one {H}x{W} image needs ~{self.plan_bytes_estimate(1, H, W, train) / 2**30:.1f} GiB "
                               f"of HBM for its launch plans ({free / 2**30:.1f} GiB available)")
        self._mb_cache[key] = mb
        return mb

    def drop_plans(self) -> None:
        """Free every module's launch plans (activation buffers) and what torch's allocator caches; they are rebuilt on demand."""
        for m in self.mods:
            m._plans.clear()
        # the TRAINING shapes keep their decision: the micro-batch - and with it the accumulation order of the gradients - stays the same
        # from epoch to epoch however often validation shapes push the plans out
        self._mb_cache = {k: v for k, v in self._mb_cache.items() if k[3]}
        torch.cuda.empty_cache()

    def check_memory(self, B: int, H: int, W: int) -> int:
        """Fail before the first batch, with the numbers, if the configuration cannot run; returns the micro-batch."""
        mb = self.micro_batch_for(B, H, W)
        if mb != B and parallel.rank() == 0:
            print(f"[uir] batch {B} at {H}x{W} runs as {B // mb} micro-batches of {mb} (gradient accumulation; the launch plans keep "
                  f"every activation of their micro-batch in HBM: ~{self.plan_bytes_estimate(mb, H, W) / 2**30:.0f} GiB)")
        return mb

    def _weights(self, mb: int, B: int) -> float:
        if self.kind in L.LOSS_ADAPTIVE_KINDS and mb != B:
            raise NotImplementedError("AdaptiveLuminanceCharbonnier cannot be accumulated over micro-batches (losses.py:135-138)")
        f = mb / B                                # see TrainStep._weights (losses the reference also divides by the batch size)
        return f * f if self.kind in (0, 1, 4) else f

    def _join_bulk_lanes(self) -> None:
        """The modules' backward lists leave their weight-gradient lanes running (TokenPlan.defer_bulk_join): ONE join per plan here,
        after the whole backward, instead of a wait for the lane's tail at each of the ~27 module boundaries."""
        for m in self.mods:
            for plan in m._plans.values():
                plan.join_bulk()

    def __call__(self, raw: torch.Tensor, ref: torch.Tensor, lr: "float | None" = None):
        debug_check_streams(raw.device)
        for m in self.mods:                       # a module whose parameters moved (e.g. .to()) re-flattens: keep the optimizer's view
            if m._flat is not self.flats[self.mods.index(m)]:
                raise RuntimeError("the network's parameters were re-allocated after the train step was built")
        B = raw.shape[0]
        mb = self.micro_batch_for(B, raw.shape[2], raw.shape[3])
        if mb == B:
            out = self.model(raw)
            res = ops.loss_sums(out.detach(), ref, self.kind)
            dout = ops.loss_grad(out.detach(), ref, self._one, self.kind, aux=res)
            out.backward(dout)
            self._join_bulk_lanes()
            loss = res[0]
        else:
            if self._acc is None:
                self._acc = torch.zeros_like(self.bucket)
                self._wdev = torch.zeros(1, dtype=torch.float32, device=self.bucket.device)
                self._loss_acc = torch.zeros(1, dtype=torch.float32, device=self.bucket.device)
            w = self._weights(mb, B)
            self._wdev.fill_(w * self.loss_scale)
            n = self.bucket.numel()
            for i, o in enumerate(range(0, B, mb)):
                out = self.model(raw[o:o + mb])
                res = ops.loss_sums(out.detach(), ref[o:o + mb], self.kind)
                dout = ops.loss_grad(out.detach(), ref[o:o + mb], self._wdev, self.kind, aux=res)
                out.backward(dout)
                self._join_bulk_lanes()
                ops.flat_accumulate(self.bucket, self._acc, n, accumulate=i > 0)
                ops.axpy_scalar(self._loss_acc, res[0], w, first=i == 0)
            ops.flat_accumulate(self._acc, self.bucket, n, accumulate=False)
            loss = self._loss_acc[0].clone()
        scale = parallel.allreduce_flat_grads(self.bucket, self.bucket.numel(), self.pg)
        norm = self.opt.step(lr=lr, grad_scale=scale / self.loss_scale)
        return loss, norm[0]

    @torch.no_grad()
    def evaluate(self, raw: torch.Tensor, ref: torch.Tensor, with_ssim: bool = False):
        B = raw.shape[0]
        mb = self.micro_batch_for(B, raw.shape[2], raw.shape[3], train=False)
        if mb == B:
            out = self.model(raw)
            res = ops.loss_sums(out, ref, self.kind)
            if with_ssim:
                return res[0], res[1], ops.ssim(ref, out, data_range=1.0)
            return res[0], res[1]
        w = self._weights(mb, B)
        loss = mse = ss = None
        for o in range(0, B, mb):
            out = self.model(raw[o:o + mb])
            res = ops.loss_sums(out, ref[o:o + mb], self.kind)
            l_i, m_i = res[0] * w, torch.pow(10.0, -res[1] / 10.0) * (mb / B)      # PSNR recombines through the mean squared error
            loss = l_i if loss is None else loss + l_i
            mse = m_i if mse is None else mse + m_i
            if with_ssim:
                s_i = ops.ssim(ref[o:o + mb], out, data_range=1.0) * (mb / B)
                ss = s_i if ss is None else ss + s_i
        psnr = -10.0 * torch.log10(mse)
        return (loss, psnr, ss) if with_ssim else (loss, psnr)


def make_train_step(model, loss_name: str, lr: float, weight_decay: float = 0.0):
    """TrainStep for the single-plan SpectralTransformer, ModuleTrainStep for a network of launch-plan modules."""
    if hasattr(model, "flat_params"):
        return TrainStep(model, loss_name, lr=lr, weight_decay=weight_decay)
    if hasattr(model, "plan_modules"):
        return ModuleTrainStep(model, loss_name, lr=lr, weight_decay=weight_decay)
    raise TypeError(f"{type(model).__name__} is not a launch-plan network of this package")
