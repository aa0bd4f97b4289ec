"""Paired raw/reference image dataset with the reference's pairing rule: the i-th file of
``sorted(os.listdir(raw_dir))`` goes with the i-th file of ``sorted(os.listdir(ref_dir))``
(src/DataManipulation/UIEBDataset.py:10-11, 24-25).  Images are decoded with PIL, resized
(bilinear) to ``image_size`` and mapped to [-1, 1] = Normalize(0.5, 0.5) of ToTensor
(src/DataManipulation/DataLoader.py:13-20), without torchvision."""
from __future__ import annotations

import os

import numpy as np
import torch
from torch.utils.data import Dataset


def load_image(path: str, image_size) -> torch.Tensor:
    from PIL import Image
    with Image.open(path) as im:
        im = im.convert("RGB")
        if image_size is not None:
            w, h = (image_size, image_size) if isinstance(image_size, int) else (image_size[1], image_size[0])
            if im.size != (w, h):
                im = im.resize((w, h), Image.BILINEAR)
        a = np.asarray(im, dtype=np.float32) / 255.0
    t = torch.from_numpy(a).permute(2, 0, 1).contiguous()
    return (t - 0.5) / 0.5


def decode_image(path: str) -> np.ndarray:
    """Decode only: packed RGB bytes [h][w][3] of the file's own size (the resize runs on the GPU, device_resize.py)."""
    from PIL import Image
    with Image.open(path) as im:
        return np.asarray(im.convert("RGB"), dtype=np.uint8).copy()


class UIEBDataset(Dataset):
    def __init__(self, raw_dir, ref_dir, image_size=1024, decode_only=False):
        self.raw_dir, self.ref_dir = raw_dir, ref_dir
        self.raw_images = sorted(os.listdir(raw_dir))
        self.ref_images = sorted(os.listdir(ref_dir))
        self.image_size = image_size
        self.decode_only = decode_only

    def __len__(self):
        return len(self.raw_images)

    def __getitem__(self, idx):
        if self.decode_only:     # uint8 tensors: DataLoader workers hand tensors over through shared memory, arrays through a pipe
            return (torch.from_numpy(decode_image(os.path.join(self.raw_dir, self.raw_images[idx]))),
                    torch.from_numpy(decode_image(os.path.join(self.ref_dir, self.ref_images[idx]))))
        raw = load_image(os.path.join(self.raw_dir, self.raw_images[idx]), self.image_size)
        ref = load_image(os.path.join(self.ref_dir, self.ref_images[idx]), self.image_size)
        return raw, ref
