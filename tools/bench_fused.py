"""The tile-fused kernels of the SpectralTransformer step alone, at the step's own sizes (B = 16): uir_gdfn_fwd (with the t1 / t2
write-out), uir_qkv_fwd, uir_attnv_fwd and uir_dwln_bwd (both layouts) - microseconds per launch and TB/s on the launch's algorithmic
bytes.  Run once per UIR_FUSED_SPLIT value to compare the split-bf16 and the f32-MFMA contractions:
    UIR_FUSED_SPLIT=0 python tools/bench_fused.py [reps=20] [filter]"""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from underwater_image_restoration_amd.engine import Recorder, hidden_of, pad16

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
only = sys.argv[2] if len(sys.argv) > 2 else ""
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
rnd = lambda *s: torch.rand(*s, device=dev, generator=g) * 2 - 1
B = 16
print(f"UIR_FUSED_SPLIT={os.environ.get('UIR_FUSED_SPLIT', '(default)')}", flush=True)


def timeit(r, label):
    if only and only not in label:
        r.fwd = []; r.cur = r.fwd; r.meta[id(r.fwd)] = []; r.lanes[id(r.fwd)] = []; r.variants[id(r.fwd)] = []
        return
    sp = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    calls = [(fn, args) for fn, args, _ in r.fwd if fn is not None][:1]      # the kernel itself, not its slab reductions
    nbytes = r.meta[id(r.fwd)][0][2]
    for _ in range(2):
        for fn, args in calls:
            assert fn(*args, sp) == 0, label
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        for fn, args in calls:
            fn(*args, sp)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1000 / reps
    print(f"{label:44s} {us:8.1f} us  {nbytes / us / 1e6:6.2f} TB/s", flush=True)
    r.fwd = []; r.cur = r.fwd; r.meta[id(r.fwd)] = []; r.lanes[id(r.fwd)] = []; r.variants[id(r.fwd)] = []      # the recorder takes the next launch


for S, Cc in ((256, 32), (256, 16), (128, 32), (64, 64), (32, 128)):
    M, dims = B * S * S, (B, S, S)
    h = hidden_of(Cc); hp = pad16(h); ldg = (2 * hp + 31) & ~31
    tag = f"C={Cc:3d} {S}x{S}"
    x, g1, b1 = rnd(M, Cc), rnd(Cc), rnd(Cc)
    r = Recorder(dev, scratch_floats=1 << 20)
    y, st = r.alloc(M, Cc), r.alloc(M, 2)
    T1, T2 = r.alloc(M, ldg)[:, :2 * hp], r.alloc(M, ldg)[:, :2 * hp]
    r.gdfn_fwd("g", x, (g1, b1), rnd(2 * h, Cc) * .2, rnd(2 * h, 9) * .3, rnd(Cc, h) * .1, y, C_=Cc, h=h, hp=hp, dims=dims, T1=T1, T2=T2, stats=st, il=1 if Cc <= 32 else 0)
    timeit(r, f"{tag} gdfn_fwd")
    if Cc <= 32:           # as the train step runs it: only the post-depthwise tensor is written
        r.gdfn_fwd("g", x, (g1, b1), rnd(2 * h, Cc) * .2, rnd(2 * h, 9) * .3, rnd(Cc, h) * .1, y, C_=Cc, h=h, hp=hp, dims=dims, T1=None, T2=T2, stats=st, il=1)
        timeit(r, f"{tag} gdfn_fwd, t2 only")
    del r, T1, T2
    r = Recorder(dev, scratch_floats=1 << 20)
    qp, q, st = r.alloc(M, 3 * Cc), r.alloc(M, 3 * Cc), r.alloc(M, 2)
    r.qkv_fwd("q", x, (g1, b1), rnd(3 * Cc, Cc) * .2, rnd(3 * Cc, 9) * .3, qp, q, C_=Cc, dims=dims, stats=st)
    timeit(r, f"{tag} qkv_fwd")
    if Cc <= 32:
        r.qkv_fwd("q", x, (g1, b1), rnd(3 * Cc, Cc) * .2, rnd(3 * Cc, 9) * .3, None, q, C_=Cc, dims=dims, stats=st)
        timeit(r, f"{tag} qkv_fwd, no pre-conv output")
    del r, qp, q
    r = Recorder(dev, scratch_floats=1 << 20)
    u, vf, y = r.alloc(M, Cc), r.alloc(M, Cc), r.alloc(M, Cc)
    r.attnv_fwd("a", x, rnd(B, Cc * Cc) * .2, rnd(B, Cc * Cc) * .2, rnd(Cc, 9) * .3, x, u, vf, y, C_=Cc, dims=dims)
    timeit(r, f"{tag} attnv_fwd")
    del r, u, vf, y
    if Cc > 32:
        continue
    for split in (True, False):
        hh, hpp, ch = (h, hp, 0) if split else (0, 0, 3 * Cc)
        cols, chreal = (2 * hp, 2 * h) if split else (ch, ch)
        r = Recorder(dev, scratch_floats=1 << 20)
        G, XP = rnd(M, cols), rnd(M, cols)
        if split:
            G[:, h:hp] = 0; G[:, hp + h:] = 0; XP[:, h:hp] = 0; XP[:, hp + h:] = 0
        dRes, stats = rnd(M, Cc), torch.stack([rnd(M) * 0.1, 1 + 0.1 * rnd(M)], 1).contiguous()
        dX, dWdw, dW1, dgb = r.alloc(M, Cc), r.alloc(chreal, 9), r.alloc(chreal, Cc), r.alloc(2 * Cc)
        r.dwln_bwd("k", G, XP, rnd(chreal, 9), rnd(chreal, Cc), x, stats, g1, b1, dRes, dX, dWdw, dW1, dgb[:Cc], dgb[Cc:], C_=Cc, dims=dims,
                   h=hh, hp=hpp, ch=ch, il=1 if split else 0)     # interleaved GDFN columns, as the step runs them
        timeit(r, f"{tag} dwln_bwd {'gdfn' if split else 'qkv'}")
        r.dwln_bwd("k", G, None, rnd(chreal, 9), rnd(chreal, Cc), x, stats, g1, b1, dRes, dX, dWdw, dW1, dgb[:Cc], dgb[Cc:], C_=Cc, dims=dims,
                   h=hh, hp=hpp, ch=ch, il=1 if split else 0)
        timeit(r, f"{tag} dwln_bwd {'gdfn' if split else 'qkv'}, XP recomputed")
        del r, G, XP
