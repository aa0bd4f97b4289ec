"""Record Pillow's own answers for the input stage (row (f)-1): run in the BUILD container, where PIL imports.

The reference resizes every decoded image with ``transforms.Resize((1024,1024), BILINEAR)`` on a PIL image
(src/DataManipulation/DataLoader.py:13-14), i.e. ``PIL.Image.resize``; this script stores what Pillow returns
for inputs produced by the integer recipe :func:`recipe_image` (regenerated on both sides, never stored), plus the
fp32 ToTensor+Normalize values for one case computed with torch exactly as torchvision does
(``img.to(float32).div(255)`` then ``sub_(mean).div_(std)``).  Output: ``tests/golden/resize_pil.npz``.
"""
import os

import numpy as np

# (in_h, in_w, out_h, out_w): up, down, mixed, one axis only, identity, degenerate, strong antialias
CASES = [(37, 53, 64, 64), (300, 200, 64, 64), (64, 64, 64, 64), (64, 50, 64, 80), (50, 64, 80, 64),
         (480, 640, 96, 96), (1, 1, 8, 8), (2, 3, 17, 5), (700, 900, 48, 40), (33, 65, 128, 128),
         (1000, 30, 20, 300), (5, 2000, 64, 64), (183, 275, 256, 256)]


def recipe_image(h: int, w: int, seed: int) -> np.ndarray:
    """uint8 [h][w][3] from an integer hash; every third case is a smooth ramp with hard edges instead."""
    y, x, c = np.meshgrid(np.arange(h, dtype=np.uint64), np.arange(w, dtype=np.uint64), np.arange(3, dtype=np.uint64),
                          indexing="ij")
    v = (y * np.uint64(131) + x * np.uint64(71) + c * np.uint64(29) + np.uint64(seed * 977 + 1)) * np.uint64(2654435761)
    v = (v >> np.uint64(13)) & np.uint64(255)
    if seed % 3 == 2:
        ramp = (y * np.uint64(3) + x * np.uint64(2) + c * np.uint64(40)) & np.uint64(255)
        v = np.where(((y // np.uint64(7)) + (x // np.uint64(5))) % np.uint64(2) == 0, ramp, np.uint64(255) - ramp)
    return v.astype(np.uint8)


def main():
    import PIL
    import torch
    from PIL import Image
    out = {"pillow_version": np.array(PIL.__version__), "cases": np.array(CASES, np.int32)}
    for i, (h, w, oh, ow) in enumerate(CASES):
        a = recipe_image(h, w, i)
        out[f"out{i}"] = np.asarray(Image.fromarray(a).resize((ow, oh), Image.BILINEAR))
    t = torch.from_numpy(out["out0"].copy()).permute(2, 0, 1).contiguous().to(torch.float32).div(255)
    mean = torch.tensor([0.5, 0.5, 0.5]).view(-1, 1, 1)
    std = torch.tensor([0.5, 0.5, 0.5]).view(-1, 1, 1)
    out["norm0"] = t.sub_(mean).div_(std).numpy()
    u = torch.arange(256, dtype=torch.uint8).to(torch.float32).div(255)
    out["lut"] = u.sub_(0.5).div_(0.5).numpy()
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "resize_pil.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
