"""The fused train step of the hot path (src/ModelTrainer.py:78-90): zero_grad -> forward ->
loss -> backward -> clip_grad_norm_(1.0) -> Adam.step, as launch lists + three reductions on
one HIP stream with no host synchronisation.  Under data parallelism (one process per GPU) the
flat gradient buffer is all-reduced once over RCCL/xGMI between backward and the optimizer.
"""
from __future__ import annotations

import os

import torch
import torch.distributed as dist

from . import _lib as L
from . import ops
from . import parallel


class TrainStep:
    def __init__(self, model, loss_name: str = "L1", lr: float = 3e-4, max_norm: float = 1.0, process_group=None):
        if loss_name not in L.LOSS_KINDS:
            raise ValueError(f"Unsupported loss: {loss_name}")
        self.model = model
        self.kind = L.LOSS_KINDS[loss_name]
        self.flat = model.flat_params()
        self.opt = ops.FusedClipAdam(self.flat, lr=lr, max_norm=max_norm)
        self.pg = process_group
        self.world = parallel.world_size(process_group)
        if self.world > 1:
            parallel.broadcast_flat_params(self.flat.flat, process_group)
        self._one = torch.ones(1, dtype=torch.float32, device=self.flat.device)
        self._graph_on = os.environ.get("UIR_GRAPH", "0") == "1" and self.flat.device.type == "cuda"
        # opt-in (UIR_GRAPH=1): on ROCm 7.2 replaying the ~1400-node graph measured 62.2 ms/step vs 57.9 eager
        self._graphs, self._eager_calls = {}, {}
        # optional (UIR_MAIN_HIPRIO=1): critical chain on a high-priority stream; measured neutral on MI355X
        self._hp = None
        if self.flat.device.type == "cuda" and os.environ.get("UIR_MAIN_HIPRIO", "0") == "1":
            hi = min(torch.cuda.Stream.priority_range()) if hasattr(torch.cuda.Stream, "priority_range") else 0
            self._hp = torch.cuda.Stream(device=self.flat.device, priority=hi)

    def __call__(self, raw: torch.Tensor, ref: torch.Tensor, lr: float | None = None):
        """One optimisation step on a device-resident batch.  Returns (loss, grad_norm) as
        device tensors (reading them is the only synchronisation, and it is the caller's)."""
        if self._hp is not None:
            cur = torch.cuda.current_stream()
            self._hp.wait_stream(cur)
            with torch.cuda.stream(self._hp):
                res, norm = self._step(raw, ref, lr)
            cur.wait_stream(self._hp)
            res.record_stream(cur)      # allocated on the high-priority stream, consumed by the caller's stream
            norm.record_stream(cur)
            return res, norm
        return self._step(raw, ref, lr)

    def _step(self, raw, ref, lr):
        plan = self.model._plan_for(raw)
        key = (id(plan), tuple(raw.shape))
        if self._graph_on and self.world == 1:
            g = self._graphs.get(key)
            if g is None:
                self._eager_calls[key] = self._eager_calls.get(key, 0) + 1
                if self._eager_calls[key] > 2:          # plan, streams and allocator pools are warm: capture
                    g = self._capture(plan, ref, key)
            if g is not None:
                plan.in_nchw.copy_(raw)
                g["ref"].copy_(ref)
                self.opt.advance(lr=lr, grad_scale=1.0)
                g["graph"].replay()
                self.model._fwd_serial += 1
                return g["res"][0], self.opt.norm[0]
        out = plan.forward(raw)
        res = ops.loss_sums(out, ref, self.kind)
        dout = ops.loss_grad(out, ref, self._one, self.kind)
        plan.backward(dout)
        self.model._fwd_serial += 1
        # single flat bucket of the live gradients (8.8 MB): one all-reduce over xGMI
        scale = parallel.allreduce_flat_grads(self.flat.grad, self.flat.n_live, self.pg)
        norm = self.opt.step(lr=lr, grad_scale=scale)
        return res[0], norm[0]

    def _capture(self, plan, ref, key):
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
                dout = ops.loss_grad(plan.out_nchw, ref_s, self._one, self.kind)
                plan.backward(dout)
                self.opt.launch()
            g = {"graph": graph, "ref": ref_s, "res": res}
        except Exception as e:   # keep training on the eager launch lists (same kernels, same results)
            import warnings
            warnings.warn(f"hipGraph capture failed ({type(e).__name__}: {e}); staying on eager launch lists")
            self._graph_on = False
            torch.cuda.synchronize()
            return None
        self._graphs[key] = g
        return g

    @torch.no_grad()
    def evaluate(self, raw: torch.Tensor, ref: torch.Tensor, with_ssim: bool = False):
        """Validation batch (src/ModelTrainer.py:134-158): -> (loss, PSNR) device tensors, plus torchSSIM(ref, out)
        (:23-24,149) with ``with_ssim`` (needs H, W >= 11)."""
        plan = self.model._plan_for(raw)
        out = plan.forward(raw)
        self.model._fwd_serial += 1
        res = ops.loss_sums(out, ref, self.kind)
        if with_ssim:
            return res[0], res[1], ops.ssim(ref, out, data_range=1.0)
        return res[0], res[1]
