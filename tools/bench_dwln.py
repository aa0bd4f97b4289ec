"""Microbenchmark of uir_dwln_bwd (csrc/dwln.hip) at the sizes of the train step: B = 16, 256 x 256 (and the level-1 size), C = 32 / 16,
GDFN and qkv layouts.  Prints microseconds per launch and the rate on the kernel's algorithmic bytes.
    python tools/bench_dwln.py [runs=20]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from underwater_image_restoration_amd.engine import Recorder, hidden_of, pad16

RUNS = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
rnd = lambda *s: torch.rand(*s, device=dev, generator=g) * 2 - 1
CASES = [(16, 256, 256, 32, True), (16, 256, 256, 32, False), (16, 256, 256, 16, True), (16, 256, 256, 16, False),
         (16, 128, 128, 32, True), (16, 128, 128, 32, False)]
if len(sys.argv) > 2 and sys.argv[2] == "first":
    CASES = CASES[:1]
for (B, H, W, C, split) in CASES:
    M = B * H * W
    h = hidden_of(C) if split else 0
    hp = pad16(h) if split else 0
    ch = 0 if split else 3 * C
    cols = 2 * hp if split else ch
    chreal = 2 * h if split else ch
    r = Recorder(dev, scratch_floats=1 << 20)
    G, XP = rnd(M, cols), rnd(M, cols)
    if split:
        G[:, h:hp] = 0; G[:, hp + h:] = 0; XP[:, h:hp] = 0; XP[:, hp + h:] = 0
    X, dRes, stats = rnd(M, C), rnd(M, C), torch.stack([rnd(M) * 0.1, 1 + 0.1 * rnd(M)], 1).contiguous()
    Wdw, W1, gam, bet = rnd(chreal, 9), rnd(chreal, C), rnd(C), rnd(C)
    dX, dWdw, dW1, dgb = r.alloc(M, C), r.alloc(chreal, 9), r.alloc(chreal, C), r.alloc(2 * C)
    r.dwln_bwd("k", G, XP, Wdw, W1, X, stats, gam, bet, dRes, dX, dWdw, dW1, dgb[:C], dgb[C:], C_=C, dims=(B, H, W), h=h, hp=hp, ch=ch)
    fn, args, _ = r.fwd[0]
    st = torch.cuda.current_stream()
    import ctypes as Cc
    sp = Cc.c_void_p(st.cuda_stream)
    for _ in range(3):
        fn(*args, sp)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(RUNS):
        fn(*args, sp)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1000 / RUNS
    nbytes = 4 * M * (2 * cols + 3 * C)
    print(f"B={B} {H}x{W} C={C} {'gdfn' if split else 'qkv '}: {us:8.1f} us  {nbytes / us / 1e3:7.1f} GB/s on {nbytes / 1e9:.2f} GB", flush=True)
