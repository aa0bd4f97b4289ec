"""Evaluation metrics of the reference on the GPU, through the C ABI only.

``uiqm``  getUIQM of the reference's uqim_utils.py:176-190 (UIQM = 0.0282 UICM + 0.2953 UISM + 3.5753 UIConM) for a batch.
PSNR / SSIM of the validation loop live in ``ops.loss_sums`` / ``ops.ssim`` (src/ModelTrainer.py:17-24).
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib as L

lib = L.lib


def uiqm(images: torch.Tensor) -> torch.Tensor:
    """images: ``[N, H, W, 3]`` on the GPU, uint8 (the bytes ``cv2.imread`` / ``PIL`` give, as eval.ipynb feeds them) or
    float32 (what the reference's ``x.astype(np.float32)`` would hold); H, W >= 10.  Returns ``[N, 4]`` float64 on the
    device: (uiqm, uicm, uism, uiconm) per image, the tuple ``getUIQM`` returns.

    Like the reference, raises ``ValueError('math domain error')`` when a 10 x 10 window has a max/min ratio <= 0 (images
    with negative values: ``math.log`` in uqim_utils.py:81,171) - checking that flag is the one host read of this call."""
    if images.device.type != "cuda":
        raise RuntimeError("uiqm: expected a GPU tensor (no CPU fallback in the HIP path)")
    if images.dim() != 4 or images.shape[3] != 3:
        raise ValueError(f"uiqm expects [N, H, W, 3] images, got {tuple(images.shape)}")
    if images.dtype not in (torch.uint8, torch.float32):
        images = images.float()
    images = images.contiguous()
    N, H, W, _ = images.shape
    nwork = lib.uir_uiqm_work_doubles(N, H, W)
    if nwork < 0:
        raise ValueError("uiqm needs N >= 1 and H, W >= 10 (one 10 x 10 window)")
    work = torch.empty(nwork, dtype=torch.float64, device=images.device)
    out = torch.empty(N, 4, dtype=torch.float64, device=images.device)
    flags = torch.empty(N, dtype=torch.int32, device=images.device)
    a = L.UiqmArgs()
    if images.dtype == torch.uint8:
        a.img_u8, a.img_f32 = images.data_ptr(), None
    else:
        a.img_f32, a.img_u8 = images.data_ptr(), None
    a.work, a.out, a.flags = work.data_ptr(), out.data_ptr(), flags.data_ptr()
    a.N, a.H, a.W = N, H, W
    L.check(lib.uir_uiqm(C.byref(a), C.c_void_p(torch.cuda.current_stream().cuda_stream)), "uir_uiqm")
    if bool(flags.any()):
        raise ValueError("math domain error")
    return out
