"""Deep-level (L2: M=65536,C=64; L3: M=16384,C=128) GEMM shapes in isolation.  usage: python tools/microbench_deep.py [reps]"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from underwater_image_restoration_amd import _lib as L
from underwater_image_restoration_amd.engine import Recorder, pad16
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
only = sys.argv[2] if len(sys.argv) > 2 else ""
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
rnd = lambda *s: torch.rand(*s, device=dev, generator=g) * 2 - 1
for lvl, (M, C) in (("L2", (65536, 64)), ("L3", (16384, 128))):
    h = int(C * 2.66); hp = pad16(h)
    r = Recorder(dev, scratch_floats=8 << 20)
    x = rnd(M, C); gp = rnd(M, 2 * hp); gp[:, h:hp] = 0; gp[:, hp + h:] = 0
    qkvb = r.alloc(M, 3 * C); y = r.alloc(M, C); gout = r.alloc(M, 2 * hp); st = r.alloc(M, 2)
    Wqkv, Win, Wout = rnd(3 * C, C) * .1, rnd(2 * h, C) * .1, rnd(C, h) * .1
    g1, b1 = rnd(C), rnd(C)
    cases = {}
    def case(name, fn):
        r.fwd = []; r.cur = r.fwd; r.meta[id(r.fwd)] = []; r.lanes[id(r.fwd)] = []
        fn(); cases[name] = (r.fwd, r.meta[id(r.fwd)])
    case("qkv_ln", lambda: r.gemm("a", x, Wqkv, qkvb, N=3 * C, K=C, w_sn=C, w_sk=1, pro=L.PRO_LN, ln=(g1, b1), stats=st))
    case("qkv_plain", lambda: r.gemm("a", x, Wqkv, qkvb, N=3 * C, K=C, w_sn=C, w_sk=1))
    case("ffn_in_plain", lambda: r.gemm("a", x, Win, gout, N=2 * hp, K=C, w_sn=C, w_sk=1, n_grp=(h, hp)))
    case("ffn_in_ln", lambda: r.gemm("a", x, Win, gout, N=2 * hp, K=C, w_sn=C, w_sk=1, n_grp=(h, hp), pro=L.PRO_LN, ln=(g1, b1), stats=st))
    case("ffn_out_gate", lambda: r.gemm("a", gp, Wout, y, N=C, K=h, w_sn=h, w_sk=1, pro=L.PRO_GATE, gate_off=hp, R=x))
    case("dgate", lambda: r.gemm("a", x, Wout, gout, N=hp, K=C, w_sn=1, w_sk=h, n_grp=(h, hp), epi=L.EPI_GATE_BWD, G=gp, g_off=hp))
    case("dt2", lambda: r.gemm("a", gp, Win, y, N=C, K=2 * hp, w_sn=1, w_sk=C, k_grp=(h, hp)))
    case("dt1", lambda: r.gemm("a", qkvb, Wqkv, y, N=C, K=3 * C, w_sn=1, w_sk=C))
    torch.cuda.synchronize()
    for name, (ops, meta) in cases.items():
        if only and only != f"{lvl}.{name}":
            continue
        r.run(ops); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps): r.run(ops)
        torch.cuda.synchronize()
        us = (time.perf_counter() - t0) / reps * 1e6
        nb = sum(m[2] for m in meta); fl = sum(m[3] for m in meta)
        print(f"{lvl} {name:14s} {us:8.1f} us  {nb / 1e6:7.1f} MB  {nb / us / 1e6:6.2f} TB/s  {fl / us / 1e6:7.2f} TFLOP/s", flush=True)
