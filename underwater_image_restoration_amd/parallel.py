"""Data parallelism for the train step: one process per GPU, replicated weights and optimizer
state, ONE all-reduce per step over the flat gradient bucket of the live parameters
(2,200,536 floats = 8.8 MB) - RCCL over xGMI on MI355X (backend "nccl" in PyTorch-ROCm), gloo on
CPU for tests.  The reference has no distributed code at all; the contract
here is SURVEY.md section 8(e): each rank runs the reference step on its own batch shard, gradients
are averaged, every rank applies the same clip + Adam update.
"""
from __future__ import annotations

import os

import torch
import torch.distributed as dist

from . import knobs


def world_size(group=None) -> int:
    return dist.get_world_size(group) if (dist.is_available() and dist.is_initialized()) else 1


def rank(group=None) -> int:
    return dist.get_rank(group) if (dist.is_available() and dist.is_initialized()) else 0


def init_from_env(device: "torch.device | None" = None) -> int:
    """Initialise torch.distributed from RANK/WORLD_SIZE/MASTER_* (torchrun).  Returns world size."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = "nccl" if (device is not None and device.type == "cuda") else "gloo"
        kw = {"device_id": device} if backend == "nccl" else {}
        dist.init_process_group(backend, **kw)
    return world


def allreduce_flat_grads(flat_grad: torch.Tensor, n_live: int, group=None) -> float:
    """Sum the live-gradient bucket over the ranks (in place); returns the scale (1/world) the
    optimizer must apply - the fused clip+Adam kernel folds it into its gradient read, so the
    bucket is touched exactly once by the collective and once by the optimizer."""
    w = world_size(group)
    if w == 1 and not (knobs.flag("UIR_FORCE_ALLREDUCE") and dist.is_initialized()):
        return 1.0
    dist.all_reduce(flat_grad[:n_live], op=dist.ReduceOp.SUM, group=group)
    return 1.0 / w


def broadcast_flat_params(flat: torch.Tensor, group=None, src: int = 0) -> None:
    """Make every rank start from rank `src`'s weights (replicated data parallelism)."""
    if world_size(group) > 1:
        dist.broadcast(flat, src=src, group=group)


def shard_indices(n: int, group=None):
    """Contiguous-stride shard of range(n) for this rank (every rank gets floor(n/world) items)."""
    w, r = world_size(group), rank(group)
    per = n // w
    return list(range(r, per * w, w))
