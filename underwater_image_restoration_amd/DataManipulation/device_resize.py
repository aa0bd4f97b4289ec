"""Input stage on the GPU: the reference's per-item ``Resize((1024,1024), BILINEAR) -> ToTensor ->
Normalize(0.5, 0.5)`` (src/DataManipulation/DataLoader.py:13-20) as one HIP launch per batch.

DataLoader workers only decode (PIL -> packed RGB bytes of the native size); the training process packs a batch's
bytes into a pinned staging buffer, copies them to the card on a side stream and launches ``uir_resize_norm_u8``
(csrc/resize.hip), which resamples exactly as Pillow does and writes the loader's fp32 ``[n,3,S,S]`` items.  Against
the CPU path this moves 3 B per *source* pixel over PCIe instead of 12 B per *output* pixel and takes the resize off
the host cores.  :class:`DevicePrefetchLoader` keeps one batch in flight behind the train step (two staging sets).

There is no CPU fallback: without the HIP library / a GPU this module raises.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, List, Sequence, Tuple

import numpy as np
import torch

from .. import _lib as L


def _axis_table(in_size: int, out_size: int) -> Tuple[int, np.ndarray, np.ndarray]:
    """(ksize, bounds[out][2], kk[out][ksize]) from the library's host function (Pillow's precompute_coeffs)."""
    ks = L.lib.uir_resize_ksize(in_size, out_size)
    if ks < 1:
        raise L.UirError(f"uir_resize_ksize({in_size}, {out_size}) failed")
    bounds = np.zeros((out_size, 2), np.int32)
    kk = np.zeros((out_size, ks), np.int32)
    if L.lib.uir_resize_coeffs(in_size, out_size, bounds.ctypes.data, kk.ctypes.data, kk.size) != ks:
        raise L.UirError(f"uir_resize_coeffs({in_size}, {out_size}) failed")
    return ks, bounds, kk


class _Staging:
    """One set of pinned host buffers + their device twins; reused every other batch."""

    def __init__(self, dev):
        self.dev = dev
        self.src_h = self.src_d = None
        self.tab_h = self.tab_d = None
        self.img_h = self.img_d = None
        self.done = None                       # event: the last launch that read this set

    def fit(self, src_bytes: int, tab_ints: int, n_img: int):
        def grow(h, d, n, dtype):
            if h is None or h.numel() < n:
                n = int(n * 1.25) + 64
                h = torch.empty(n, dtype=dtype).pin_memory()
                d = torch.empty(n, dtype=dtype, device=self.dev)
            return h, d
        self.src_h, self.src_d = grow(self.src_h, self.src_d, src_bytes, torch.uint8)
        self.tab_h, self.tab_d = grow(self.tab_h, self.tab_d, tab_ints, torch.int32)
        self.img_h, self.img_d = grow(self.img_h, self.img_d, n_img * C.sizeof(L.ResizeImage), torch.uint8)


class DeviceResizer:
    """Batch of decoded RGB images (uint8 [h_i][w_i][3], any sizes) -> fp32 [n,3,S_h,S_w] on ``device``."""

    def __init__(self, size, device="cuda"):
        self.out_h, self.out_w = (size, size) if isinstance(size, int) else (int(size[0]), int(size[1]))
        self.dev = torch.device(device)
        if self.dev.type != "cuda":
            raise RuntimeError("DeviceResizer needs the GPU; there is no CPU fallback")
        if self.dev.index is None:
            self.dev = torch.device("cuda", torch.cuda.current_device())
        self._axis: Dict[Tuple[int, int], Tuple[int, np.ndarray, np.ndarray]] = {}
        # byte -> fp32 exactly as torchvision: ToTensor = uint8.to(float32).div(255); Normalize = sub(0.5).div(0.5)
        self.lut = torch.arange(256, dtype=torch.uint8).to(torch.float32).div(255).sub_(0.5).div_(0.5).to(self.dev)
        self.sets = [_Staging(self.dev), _Staging(self.dev)]
        self.turn = 0
        self.stream = torch.cuda.Stream(device=self.dev)

    def _table(self, in_size, out_size):
        key = (in_size, out_size)
        if key not in self._axis:
            self._axis[key] = _axis_table(in_size, out_size)
        return self._axis[key]

    def __call__(self, images: Sequence[np.ndarray], want_u8: bool = False, stream=None):
        """Enqueue copy + resize of a list of uint8 [h][w][3] images on ``stream`` (default: the resizer's side stream).
        Returns (out, event) -- the consumer's stream must wait on ``event`` and, if it is another stream, call
        ``out.record_stream(consumer_stream)`` -- or (out, out_u8, event) with ``want_u8``."""
        images = [a.numpy() if torch.is_tensor(a) else a for a in images]
        if len(images) == 0:
            raise ValueError("empty batch")
        for a in images:
            if a.dtype != np.uint8 or a.ndim != 3 or a.shape[2] != 3 or a.shape[0] < 1 or a.shape[1] < 1:
                raise ValueError("images must be uint8 [h][w][3]")
        shapes = [(a.shape[0], a.shape[1]) for a in images]

        def fill(src_np, offsets):
            for a, o in zip(images, offsets):
                src_np[o:o + a.size] = np.ascontiguousarray(a).reshape(-1)
        return self._submit(shapes, fill, want_u8, stream)

    def submit_packed(self, packed: torch.Tensor, shapes, want_u8: bool = False, stream=None):
        """Same, for a batch a DataLoader worker has already packed (``collate_packed``): ``packed`` = the images' bytes back
        to back (1-D uint8, shared memory), ``shapes`` = their (h, w): one host copy into the pinned staging buffer."""
        shapes = [(int(h), int(w)) for h, w in (shapes.tolist() if torch.is_tensor(shapes) else shapes)]
        if len(shapes) == 0 or packed.dtype != torch.uint8 or packed.numel() != sum(3 * h * w for h, w in shapes):
            raise ValueError("packed batch does not match its shapes")

        def fill(src_np, offsets):
            src_np[:packed.numel()] = packed.numpy()
        return self._submit(shapes, fill, want_u8, stream)

    def _submit(self, shapes, fill, want_u8, stream):
        n = len(shapes)
        st = self.sets[self.turn]
        self.turn ^= 1
        # plan: tables de-duplicated per (source extent, axis), descriptors, byte offsets
        tabs: List[np.ndarray] = []
        off: Dict[Tuple[str, int], Tuple[int, int, int]] = {}
        cur = 0
        descs = (L.ResizeImage * n)()
        src_off = 0
        for i, (h, w) in enumerate(shapes):
            if h < 1 or w < 1:
                raise ValueError("images must be uint8 [h][w][3]")
            for axis, in_size, out_size in (("h", w, self.out_w), ("v", h, self.out_h)):
                if (axis, in_size) not in off:
                    ks, b, k = self._table(in_size, out_size)
                    off[(axis, in_size)] = (cur, cur + b.size, ks)
                    tabs += [b.ravel(), k.ravel()]
                    cur += b.size + k.size
            d = descs[i]
            d.src_off, d.in_h, d.in_w = src_off, h, w
            d.hb, d.hk, d.kh = off[("h", w)]
            d.vb, d.vk, d.kv = off[("v", h)]
            src_off += h * w * 3
        if st.done is not None:
            st.done.synchronize()                 # the launch that last read this staging set has finished
        st.fit(src_off + L.RESIZE_SRC_SLACK, cur, n)
        fill(st.src_h.numpy(), [descs[i].src_off for i in range(n)])
        st.tab_h.numpy()[:cur] = np.concatenate(tabs)
        C.memmove(st.img_h.data_ptr(), descs, C.sizeof(descs))

        stream = stream or self.stream
        with torch.cuda.stream(stream):
            # allocated UNDER the launch stream: the block comes from that stream's pool of the caching allocator, so it
            # cannot be memory the train step freed on its own stream while kernels that still use it are queued there
            # (the consumer's record_stream only covers the hand-over and the later free, not this first use)
            out = torch.empty(n, 3, self.out_h, self.out_w, dtype=torch.float32, device=self.dev)
            out_u8 = torch.empty(n, self.out_h, self.out_w, 3, dtype=torch.uint8, device=self.dev) if want_u8 else None
            st.src_d[:src_off].copy_(st.src_h[:src_off], non_blocking=True)
            st.tab_d[:cur].copy_(st.tab_h[:cur], non_blocking=True)
            st.img_d[:C.sizeof(descs)].copy_(st.img_h[:C.sizeof(descs)], non_blocking=True)
            a_ = L.ResizeArgs()
            a_.src, a_.src_bytes, a_.src_capacity = st.src_d.data_ptr(), src_off, st.src_d.numel()
            a_.img, a_.img_host = st.img_d.data_ptr(), st.img_h.data_ptr()
            a_.tab, a_.tab_host, a_.tab_ints = st.tab_d.data_ptr(), st.tab_h.data_ptr(), cur
            a_.lut = self.lut.data_ptr()
            a_.out_f32 = out.data_ptr()
            a_.out_u8 = out_u8.data_ptr() if want_u8 else None
            a_.n_img, a_.out_h, a_.out_w = n, self.out_h, self.out_w
            L.check(L.lib.uir_resize_norm_u8(C.byref(a_), stream.cuda_stream), "uir_resize_norm_u8")
            ev = torch.cuda.Event()
            ev.record(stream)
        st.done = ev
        if want_u8:
            return out, out_u8, ev
        return out, ev


def collate_packed(items):
    """DataLoader collate for decode-only items, run in the worker: the batch's bytes back to back in ONE uint8 tensor
    (raw images first, then ref) plus the (h, w) of each image -- the training process maps one shared-memory block and
    does one copy per batch instead of 2*B of each."""
    imgs = [r for r, _ in items] + [f for _, f in items]
    shapes = torch.tensor([[t.shape[0], t.shape[1]] for t in imgs], dtype=torch.int32)
    packed = torch.cat([t.reshape(-1) for t in imgs])
    return packed, shapes, len(items)


class DevicePrefetchLoader:
    """Wraps a decode-only DataLoader: yields (raw, ref) fp32 ``[B,3,S,S]`` device tensors, resized on the GPU.  A
    background thread takes the packed batches from the workers, copies them into the pinned staging buffer and enqueues
    H2D + resize on the resizer's side stream, up to two batches ahead of the consumer: the ~18 ms of host work per
    batch (mapping the shared-memory block, the 40 MB copy; both release the GIL) overlaps the train step's launch loop."""

    def __init__(self, loader, size, device="cuda", ahead: int = 2):
        self.loader = loader
        self.resizer = DeviceResizer(size, device)
        self.ahead = max(1, int(ahead))

    def __len__(self):
        return len(self.loader)

    @property
    def dataset(self):
        return self.loader.dataset

    def _submit(self, batch):
        packed, shapes, nb = batch
        out, ev = self.resizer.submit_packed(packed, shapes)
        return out, ev, int(nb)

    def __iter__(self):
        import queue
        import threading
        dev = self.resizer.dev
        q: "queue.Queue" = queue.Queue(maxsize=self.ahead)
        stop = threading.Event()

        def producer():
            try:
                torch.cuda.set_device(dev)
                for batch in self.loader:
                    if stop.is_set():
                        return
                    q.put(self._submit(batch))
                q.put(None)
            except BaseException as e:          # surfaces in the consumer
                q.put(e)

        th = threading.Thread(target=producer, name="uir-input-stage", daemon=True)
        th.start()
        try:
            while True:
                item = q.get()
                if item is None:
                    break
                if isinstance(item, BaseException):
                    raise item
                out, ev, nb = item
                cur = torch.cuda.current_stream(dev)
                cur.wait_event(ev)
                out.record_stream(cur)
                yield out[:nb], out[nb:]
        finally:
            stop.set()
            while th.is_alive():                # unblock a producer waiting on a full queue, then let it finish
                try:
                    q.get_nowait()
                except queue.Empty:
                    pass
                th.join(timeout=0.05)
