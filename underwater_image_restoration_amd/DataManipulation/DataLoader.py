"""``get_dataloaders`` with the reference's signature (src/DataManipulation/DataLoader.py:11-32):
train loader shuffled, test loader not, items (raw, ref) float32 [3,S,S] in [-1,1].  The reference
hard-codes S = 1024 (:14); ``image_size`` makes that explicit (default 1024 = reference) so the
256x256 benchmark configuration can be fed from files too.  Under torch.distributed each rank
reads a disjoint shard (DistributedSampler)."""
from __future__ import annotations

import torch
from torch.utils.data import DataLoader
from torch.utils.data.distributed import DistributedSampler

from .UIEBDataset import UIEBDataset


def get_dataloaders(raw_dir, ref_dir, test_raw_dir, test_ref_dir, batch_size=16, num_workers=4, image_size=1024,
                    shuffle=True, pin_memory=True):
    train_ds = UIEBDataset(raw_dir, ref_dir, image_size)
    test_ds = UIEBDataset(test_raw_dir, test_ref_dir, image_size)
    sampler = None
    if torch.distributed.is_available() and torch.distributed.is_initialized() and torch.distributed.get_world_size() > 1:
        sampler = DistributedSampler(train_ds, shuffle=shuffle, drop_last=True)
    train_loader = DataLoader(train_ds, batch_size=batch_size, shuffle=(shuffle and sampler is None), sampler=sampler,
                              num_workers=num_workers, pin_memory=pin_memory, drop_last=sampler is not None)
    test_loader = DataLoader(test_ds, batch_size=batch_size, shuffle=False, num_workers=num_workers, pin_memory=pin_memory)
    return train_loader, test_loader
