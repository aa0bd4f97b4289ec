"""Deep-level shapes of the SpectralTransformer step in isolation (B = 16, 256 x 256 input: L2 = 64 x 64 pixels x 64 channels,
L3 = 32 x 32 x 128): every candidate launch of a TransformerBlock, hot loop, microseconds per launch.
    python tools/bench_deep.py [reps=30] [filter]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import torch
from underwater_image_restoration_amd import _lib as L
from underwater_image_restoration_amd.engine import Recorder, hidden_of, pad16
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
only = sys.argv[2] if len(sys.argv) > 2 else ""
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
rnd = lambda *s: torch.rand(*s, device=dev, generator=g) * 2 - 1
B = 16


def timeit(r, label, flops):
    ops = r.fwd
    if only and only not in label:
        r.fwd = []; r.cur = r.fwd; r.meta[id(r.fwd)] = []; r.lanes[id(r.fwd)] = []; r.variants[id(r.fwd)] = []
        return
    sp = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    calls = [(fn, args) for fn, args, _ in ops if fn is not None]
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
    print(f"{label:44s} {len(calls)} launch(es) {us:8.1f} us  {flops / us / 1e6:6.1f} TFLOP/s", flush=True)
    r.fwd = []; r.cur = r.fwd; r.meta[id(r.fwd)] = []; r.lanes[id(r.fwd)] = []; r.variants[id(r.fwd)] = []


for lvl, (S, Cc, heads) in (("L2", (64, 64, 4)), ("L3", (32, 128, 8))):
    M = B * S * S
    HW = S * S
    dims = (B, S, S)
    h = hidden_of(Cc); hp = pad16(h); ldg = (2 * hp + 31) & ~31
    r = Recorder(dev, scratch_floats=32 << 20)
    x, y, st = rnd(M, Cc), r.alloc(M, Cc), r.alloc(M, 2)
    qkvp, qkv = r.alloc(M, 3 * Cc), rnd(M, 3 * Cc)
    gp = rnd(M, ldg)[:, :2 * hp]; gp[:, h:hp] = 0; gp[:, hp + h:] = 0
    gout = r.alloc(M, ldg)[:, :2 * hp]
    Wqkv, Wdw, Win, Wffdw, Wout = rnd(3 * Cc, Cc) * .1, rnd(3 * Cc, 9) * .1, rnd(2 * h, Cc) * .1, rnd(2 * h, 9) * .1, rnd(Cc, h) * .1
    g1, b1 = rnd(Cc), rnd(Cc)
    WA, Mqk = rnd(B, Cc * Cc) * .1, rnd(B, 4 * Cc * Cc) * .1
    # ---- forward candidates
    r.qkv_fwd("q", x, (g1, b1), Wqkv, Wdw, qkvp, qkv, C_=Cc, dims=dims, stats=st)
    timeit(r, f"{lvl} fwd qkv fused (uir_qkv_fwd)", 2 * M * Cc * 3 * Cc)
    r.gemm("q", x, Wqkv, qkvp, N=3 * Cc, K=Cc, w_sn=Cc, w_sk=1, pro=L.PRO_LN, ln=(g1, b1), stats=st)
    r.dwconv("qc", qkvp, Wdw, qkv, C_=3 * Cc, dims=dims)
    timeit(r, f"{lvl} fwd qkv = LN-GEMM + dwconv", 2 * M * Cc * 3 * Cc)
    r.gdfn_fwd("g", x, (g1, b1), Win, Wffdw, Wout, y, C_=Cc, h=h, hp=hp, dims=dims, T1=gout, T2=gp, stats=st)
    timeit(r, f"{lvl} fwd gdfn fused (uir_gdfn_fwd)", 2 * M * (2 * h * Cc + h * Cc))
    gpre = r.alloc(M, ldg)[:, :2 * hp]
    r.gemm("fi", x, Win, gpre, N=2 * hp, K=Cc, w_sn=Cc, w_sk=1, n_grp=(h, hp), pro=L.PRO_LN, ln=(g1, b1), stats=st)
    r.dwconv("fc", gpre, Wffdw, gout, C_=2 * hp, dims=dims, grp=(h, hp))
    r.gemm("fo", gout, Wout, y, N=Cc, K=h, w_sn=h, w_sk=1, pro=L.PRO_GATE, gate_off=hp, R=x)
    timeit(r, f"{lvl} fwd gdfn = LN-GEMM + dwconv + gate-GEMM", 2 * M * (2 * h * Cc + h * Cc))
    # ---- backward GEMMs
    r.gemm("dt1", qkv, Wqkv, y, N=Cc, K=3 * Cc, w_sn=1, w_sk=Cc)
    timeit(r, f"{lvl} bwd dt1 (K=3C, N=C)", 2 * M * 3 * Cc * Cc)
    r.gemm("dt2", gp, Win, y, N=Cc, K=2 * hp, w_sn=1, w_sk=Cc, k_grp=(h, hp))
    timeit(r, f"{lvl} bwd dt2 one launch (k_grp)", 2 * M * 2 * h * Cc)
    r.gemm("dt2a", gp[:, :h], Win[:h], y, N=Cc, K=h, w_sn=1, w_sk=Cc)
    r.gemm("dt2b", gp[:, hp:hp + h], Win[h:], y, N=Cc, K=h, w_sn=1, w_sk=Cc, R=y)
    timeit(r, f"{lvl} bwd dt2 as x1 half + x2 half (accumulate)", 2 * M * 2 * h * Cc)
    r.gemm("dgate", x, Wout, gout, N=hp, K=Cc, w_sn=1, w_sk=h, n_grp=(h, hp), epi=L.EPI_GATE_BWD, G=gp, g_off=hp)
    timeit(r, f"{lvl} bwd dgate (gate-bwd epilogue)", 2 * M * h * Cc)
    r.gemm("dqk", qkv[:, :2 * Cc], Mqk, qkvp[:, :2 * Cc], N=2 * Cc, K=2 * Cc, w_sn=2 * Cc, w_sk=1, w_bs=4 * Cc * Cc, batches=B)
    timeit(r, f"{lvl} bwd dqk (per-sample 2C x 2C)", 2 * M * 4 * Cc * Cc)
    r.gemm("WA", qkv[:, 2 * Cc:], WA, y, N=Cc, K=Cc, w_sn=Cc, w_sk=1, w_bs=Cc * Cc, batches=B)
    timeit(r, f"{lvl} per-sample C x C GEMM (WA / WB / dv / dvf)", 2 * M * Cc * Cc)
    # UpSample's 1x1 MLPs (amp_fuse / pha_fuse, SpectralTransformer.py:165-168) run on the level's pixels too
    Wf, bf = rnd(Cc, Cc) * .1, rnd(Cc)
    hm = r.alloc(M, Cc)
    r.gemm("f0", x, Wf, hm, N=Cc, K=Cc, w_sn=Cc, w_sk=1, bias=bf, act=L.ACT_LEAKY)
    timeit(r, f"{lvl} ups 1x1 + bias + LeakyReLU (C x C)", 2 * M * Cc * Cc)
    r.gemm("f2dx", x, Wf, y, N=Cc, K=Cc, w_sn=1, w_sk=Cc, epi=L.EPI_LEAKY_BWD, G=hm)
    timeit(r, f"{lvl} ups 1x1 input gradient + LeakyReLU bwd", 2 * M * Cc * Cc)
    r.gemm("f0dx", x, Wf, y, N=Cc, K=Cc, w_sn=1, w_sk=Cc)
    timeit(r, f"{lvl} ups 1x1 input gradient (plain, C x C)", 2 * M * Cc * Cc)
    dW = r.alloc(2 * h, Cc)
    r.gram("dWin", gp, x, dW, N=2 * hp, K=Cc, o_sn=Cc, o_sk=1, n_grp=(h, hp), pro_b=L.PRO_LN, ln=(g1, b1), stats=st)
    r.flush_reductions("t")
    timeit(r, f"{lvl} bwd dWin Gram (+ reduce)", 2 * M * 2 * h * Cc)
    dWq = r.alloc(3 * Cc, Cc)
    r.gram("dWqkv", qkv, x, dWq, N=3 * Cc, K=Cc, o_sn=Cc, o_sk=1, pro_b=L.PRO_LN, ln=(g1, b1), stats=st)
    r.flush_reductions("t")
    timeit(r, f"{lvl} bwd dWqkv Gram (+ reduce)", 2 * M * 3 * Cc * Cc)
    dWo = r.alloc(Cc, h)
    r.gram("dWout", x, gp, dWo, N=Cc, K=h, o_sn=h, o_sk=1, pro_b=L.PRO_GATE, gate_off=hp)
    r.flush_reductions("t")
    timeit(r, f"{lvl} bwd dWout Gram (+ reduce)", 2 * M * h * Cc)
    dWf = r.alloc(Cc, Cc)
    r.gram("fuse.dW", x, y, dWf, N=Cc, K=Cc, o_sn=Cc, o_sk=1)
    r.flush_reductions("t")
    timeit(r, f"{lvl} ups 1x1 weight gradient (C x C) (+ reduce)", 2 * M * Cc * Cc)
    dWB = r.alloc(B, Cc * Cc)
    r.gram("dWB", x, y, dWB, N=Cc, K=Cc, o_sn=Cc, o_sk=1, o_bs=Cc * Cc, batches=B, out_elems=B * Cc * Cc)
    r.flush_reductions("t")
    timeit(r, f"{lvl} bwd per-sample dW_B Gram (+ reduce)", 2 * M * Cc * Cc)
