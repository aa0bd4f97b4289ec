"""Per-phase cycle breakdown of uir_dwln_bwd / uir_gdfn_fwd from the DIAGNOSTIC builds (csrc/uir_stamps.h, -DUIR_STAMPS):
    tools/probes/stamps/build.sh && python tools/probes/stamps/run.py
Every wave sums s_memtime deltas per phase; printed: mean over waves of each phase's share of the wave's lifetime, per wave id."""
import ctypes as C
import os
import sys
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(HERE))))
import numpy as np
import torch
from underwater_image_restoration_amd import _lib as L
from underwater_image_restoration_amd.engine import Recorder, hidden_of, pad16

dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
rnd = lambda *s: torch.rand(*s, device=dev, generator=g) * 2 - 1
NST = 12


def time_only(libpath, sym, fnargs, label, reps=10):
    lib = C.CDLL(libpath)
    fn = getattr(lib, sym)
    fn.restype = C.c_int
    sp = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    for _ in range(3):
        assert fn(*fnargs, sp) == 0
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn(*fnargs, sp)
    e1.record()
    torch.cuda.synchronize()
    print(f"  {label:52s} {e0.elapsed_time(e1) * 1000 / reps:8.1f} us")


def variants(kind, sym, fnargs):
    time_only(L.LIB_PATH if hasattr(L, "LIB_PATH") else os.path.join(os.path.dirname(L.__file__), "libuir_hip.so"), sym, fnargs, "product library")
    time_only(os.path.join(HERE, f"libstamps_{kind}.so"), sym, fnargs, "stamps (s_memtime + lgkmcnt(0) per phase)")
    time_only(os.path.join(HERE, f"libfence2_{kind}.so"), sym, fnargs, "sched_barrier at every stamp site")
    time_only(os.path.join(HERE, f"libfence3_{kind}.so"), sym, fnargs, "s_waitcnt lgkmcnt(0) at every stamp site")


def run(libname, sym, fnargs, nblocks, names):
    lib = C.CDLL(os.path.join(HERE, libname))
    fn = getattr(lib, sym)
    fn.restype = C.c_int
    sp = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    for _ in range(3):
        assert fn(*fnargs, sp) == 0
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    assert fn(*fnargs, sp) == 0
    e1.record()
    torch.cuda.synchronize()
    n = min(nblocks, 1024) * 4 * NST
    host = (C.c_uint * n)()
    assert lib.uir_stamps_fetch(host, n) == 0
    a = np.frombuffer(host, dtype=np.uint32).reshape(-1, 4, NST).astype(np.float64)
    tot = a.sum(2)
    print(f"  kernel {e0.elapsed_time(e1) * 1000:.1f} us (instrumented); wave lifetime mean {tot.mean():.0f} cycles "
          f"(min {tot.min():.0f} max {tot.max():.0f})")
    for w in range(4):
        sh = a[:, w, :].mean(0)
        print(f"  wave {w}: " + "  ".join(f"{nm} {v / sh.sum() * 100:4.1f}%" for nm, v in zip(names, sh) if nm))
    sh = a.mean((0, 1))
    print("  all   : " + "  ".join(f"{nm} {v:8.0f}" for nm, v in zip(names, sh) if nm), "(cycles per wave)")


B, S, Cc = 16, 256, 32
M, dims = B * S * S, (B, S, S)
h = hidden_of(Cc); hp = pad16(h); ldg = (2 * hp + 31) & ~31
x, g1, b1 = rnd(M, Cc), rnd(Cc), rnd(Cc)
for split in (True, False):
    hh, hpp, ch = (h, hp, 0) if split else (0, 0, 3 * Cc)
    cols, chreal = (2 * hp, 2 * h) if split else (ch, ch)
    r = Recorder(dev, scratch_floats=1 << 20)
    G, XP = rnd(M, cols), rnd(M, cols)
    dRes, stats = rnd(M, Cc), torch.stack([rnd(M) * 0.1, 1 + 0.1 * rnd(M)], 1).contiguous()
    dX, dWdw, dW1, dgb = r.alloc(M, Cc), r.alloc(chreal, 9), r.alloc(chreal, Cc), r.alloc(2 * Cc)
    # the train step's form (round 5): XP = None (centre values recomputed), the GDFN tensors chunk-interleaved; STAMPS_XP=1: the stored-XP form
    r.dwln_bwd("k", G, XP if os.environ.get("STAMPS_XP") == "1" else None, rnd(chreal, 9), rnd(chreal, Cc), x, stats, g1, b1, dRes, dX, dWdw, dW1, dgb[:Cc], dgb[Cc:],
               C_=Cc, dims=dims, h=hh, hp=hpp, ch=ch, il=1 if split else 0)
    _, args, _ = r.fwd[0]
    print(f"uir_dwln_bwd C={Cc} {S}x{S} {'gdfn' if split else 'qkv'} layout")
    run("libstamps_dwln.so", "uir_dwln_bwd", args, 512,
        ["xhat", "(a)G->LDS+bar", "w1v+load_g", "walk", "barrier", "DWA", "(c)", "(d)", "epilogue", "xp/wf-issue", "dpp+WR", ""])
    variants("dwln", "uir_dwln_bwd", args)
    del r, G, XP
r = Recorder(dev, scratch_floats=1 << 20)
y, st = r.alloc(M, Cc), r.alloc(M, 2)
T1, T2 = r.alloc(M, ldg)[:, :2 * hp], r.alloc(M, ldg)[:, :2 * hp]
r.gdfn_fwd("g", x, (g1, b1), rnd(2 * h, Cc) * .2, rnd(2 * h, 9) * .3, rnd(Cc, h) * .1, y, C_=Cc, h=h, hp=hp, dims=dims,
           T1=T1 if os.environ.get("STAMPS_XP") == "1" else None, T2=T2, stats=st, il=1)
_, args, _ = r.fwd[0]
print(f"uir_gdfn_fwd C={Cc} {S}x{S} (UIR_FUSED_SPLIT={os.environ.get('UIR_FUSED_SPLIT', '1')})")
run("libstamps_gdfn.so", "uir_gdfn_fwd", args, 1024,
    ["phase0", "(a)T1s-writes", "bar_a", "pin+T1st", "walk", "bar_b", "T2st", "(c)", "epi", "wload-issue", "(a)split+mfma", ""])
variants("gdfn", "uir_gdfn_fwd", args)
