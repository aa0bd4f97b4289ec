"""Record the reference's own UIQM outputs as a fixture.  Run in the build container only: imports
/root/reference/uqim_utils.py by file path (numpy / scipy / PIL, all present) and calls getUIQM (:176-190) and its three
parts on deterministic images.  Stored: the images (small) and the float64 results; never the reference's source.

    python tests/golden/make_uiqm_vectors.py   ->  tests/golden/uiqm_ref.npz
"""
import importlib.util
import os
import warnings

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
spec = importlib.util.spec_from_file_location("ref_uqim", "/root/reference/uqim_utils.py")
ref = importlib.util.module_from_spec(spec)
spec.loader.exec_module(ref)


def smooth(h, w, seed):
    """A natural-image-like uint8 picture: low-frequency colour fields + texture."""
    rng = np.random.RandomState(seed)
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float64)
    img = np.zeros((h, w, 3))
    for c in range(3):
        for _ in range(4):
            fx, fy, ph = rng.uniform(0.01, 0.2, 2).tolist() + [rng.uniform(0, 6.28)]
            img[:, :, c] += rng.uniform(10, 50) * np.sin(fx * xx + fy * yy + ph)
        img[:, :, c] += rng.uniform(80, 160) + rng.normal(0, 6, (h, w))
    return np.clip(np.round(img), 0, 255).astype(np.uint8)


cases = {}
rng = np.random.RandomState(7)
cases["u8_rand_64"] = rng.randint(0, 256, (64, 64, 3)).astype(np.uint8)
cases["u8_smooth_96x128"] = smooth(96, 128, 1)
cases["u8_smooth_ragged_53x37"] = smooth(53, 37, 2)                       # not multiples of the 10 x 10 window
cases["u8_smooth_256"] = smooth(256, 256, 3)                              # the size eval.ipynb uses (cv2.resize to 256 x 256)
dark = smooth(80, 70, 4)
dark[:30, :40] = 0                                                        # zero blocks: the min == 0 / top == 0 branches
cases["u8_zero_blocks_80x70"] = dark
flat = smooth(40, 50, 5)
flat[:, :, 1] = 77                                                        # constant channel: sobel max 0 -> NaN (kept as the reference has it)
cases["u8_const_channel_40x50"] = flat
cases["f32_unit_range_48x60"] = (smooth(48, 60, 6).astype(np.float32) / 255.0)          # [0,1] floats (non-integer values)
cases["f32_signed_30x41"] = rng.uniform(-1, 1, (30, 41, 3)).astype(np.float32)          # negative values: [-1,1] network range
cases["u8_ties_64x80"] = (rng.randint(0, 4, (64, 80, 3)) * 60).astype(np.uint8)         # heavy ties at the trimming thresholds
cases["u8_sat_50x50"] = np.where(rng.rand(50, 50, 3) > 0.5, 255, 0).astype(np.uint8)

out = {}
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    for name, img in cases.items():
        x = img.astype(np.float32)
        out[name + ".img"] = img
        try:
            res = ref.getUIQM(img)
        except ValueError as e:            # math.log of a negative max/min ratio: the reference's behaviour for signed images
            out[name + ".raises"] = np.array(str(e))
            print(name, "raises ValueError:", e)
            continue
        rg = x[:, :, 0].flatten() - x[:, :, 1].flatten()
        out[name + ".uiqm"] = np.array([float(v) for v in res], dtype=np.float64)            # (uiqm, uicm, uism, uiconm)
        out[name + ".mu_a_rg"] = np.float64(ref.mu_a(rg))
        out[name + ".sobel_r"] = np.asarray(ref.sobel(x[:, :, 0]), dtype=np.float32)        # scaled magnitude of channel 0
        out[name + ".eme_r"] = np.float64(ref.eme(np.multiply(ref.sobel(x[:, :, 0]), x[:, :, 0]), 10))
        print(name, img.shape, img.dtype, out[name + ".uiqm"])
np.savez_compressed(os.path.join(HERE, "uiqm_ref.npz"), **out)
print("wrote uiqm_ref.npz", os.path.getsize(os.path.join(HERE, "uiqm_ref.npz")) // 1024, "KB")
