"""``get_dataloaders`` with the reference's signature (src/DataManipulation/DataLoader.py:11-32):
train loader shuffled, test loader not, items (raw, ref) float32 [3,S,S] in [-1,1].  The reference
hard-codes S = 1024 (:14); ``image_size`` makes that explicit (default 1024 = reference) so the
256x256 benchmark configuration can be fed from files too.  Under torch.distributed each rank
reads a disjoint shard (DistributedSampler).

``device_resize=True`` keeps only the decode in the worker processes and runs the reference's
Resize -> ToTensor -> Normalize (:13-20) as one HIP launch per batch on ``device`` (device_resize.py): the loaders
then yield device tensors, bit-identical to the CPU items."""
from __future__ import annotations

import torch
from torch.utils.data import DataLoader
from torch.utils.data.distributed import DistributedSampler

from .UIEBDataset import UIEBDataset


class RankBatches:
    """batch_sampler of the validation loader under data parallelism: the reference's consecutive test batches
    (DataLoader.py:30: shuffle=False, batch i = items [i * bs, (i + 1) * bs)), of which this rank gets i = rank, rank + world, ...
    Nothing is decoded or resized for the other ranks' batches; the per-batch means recombine exactly (ModelTrainer.validate)."""
    rank_sharded = True

    def __init__(self, n_items: int, batch_size: int, rank: int, world: int):
        self.batches = [list(range(i, min(i + batch_size, n_items))) for i in range(0, n_items, batch_size)][rank::world]

    def __iter__(self):
        return iter(self.batches)

    def __len__(self):
        return len(self.batches)


def get_dataloaders(raw_dir, ref_dir, test_raw_dir, test_ref_dir, batch_size=16, num_workers=4, image_size=1024,
                    shuffle=True, pin_memory=True, device_resize=False, device="cuda", persistent_workers=None):
    train_ds = UIEBDataset(raw_dir, ref_dir, image_size, decode_only=device_resize)
    test_ds = UIEBDataset(test_raw_dir, test_ref_dir, image_size, decode_only=device_resize)
    sampler = None
    if torch.distributed.is_available() and torch.distributed.is_initialized() and torch.distributed.get_world_size() > 1:
        sampler = DistributedSampler(train_ds, shuffle=shuffle, drop_last=True)
    # worker processes are kept across epochs by default: forking a process that holds a GPU context costs ~0.3 s per worker
    # (measured: 8 workers re-forked per epoch capped either loader at ~50 images/s); the reference's DataLoader call
    # (DataLoader.py:29-30) re-forks every epoch, which changes timing only, not the items
    if persistent_workers is None:
        persistent_workers = num_workers > 0
    extra = {"persistent_workers": persistent_workers} if num_workers > 0 else {}
    if device_resize:
        from .device_resize import DevicePrefetchLoader, collate_packed
        extra["collate_fn"] = collate_packed
        pin_memory = False                 # items are variable-size byte arrays; the resizer owns the pinned staging
    train_loader = DataLoader(train_ds, batch_size=batch_size, shuffle=(shuffle and sampler is None), sampler=sampler,
                              num_workers=num_workers, pin_memory=pin_memory, drop_last=sampler is not None, **extra)
    if sampler is not None:            # data parallel: each rank loads only the validation batches it evaluates
        test_loader = DataLoader(test_ds, batch_sampler=RankBatches(len(test_ds), batch_size, torch.distributed.get_rank(),
                                                                    torch.distributed.get_world_size()),
                                 num_workers=num_workers, pin_memory=pin_memory, **extra)
    else:
        test_loader = DataLoader(test_ds, batch_size=batch_size, shuffle=False, num_workers=num_workers, pin_memory=pin_memory,
                                 **extra)
    if device_resize:
        train_loader = DevicePrefetchLoader(train_loader, image_size, device)
        test_loader = DevicePrefetchLoader(test_loader, image_size, device)
    return train_loader, test_loader
