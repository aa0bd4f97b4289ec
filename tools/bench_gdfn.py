"""Time the tile-fused GDFN forward kernel alone (store / no-store) against the three-kernel sequence."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from underwater_image_restoration_amd.engine import Recorder
from underwater_image_restoration_amd import _lib as L

DEV = "cuda:0"


def timeit(r, n=int(os.environ.get('REPS', 20))):
    r.run(r.fwd)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        r.run(r.fwd)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def main():
    B = int(os.environ.get("B", 16))
    only = os.environ.get("ONLY", "")
    for C, S in ((32, 256), (16, 256), (32, 128), (64, 64), (128, 32)):
        if only and only != f"{C}x{S}":
            continue
        h = int(C * 2.66); hp = (h + 15) & ~15; ldt = (2 * hp + 31) & ~31
        M = B * S * S
        g = torch.Generator(device="cpu").manual_seed(0)
        x = torch.rand(M, C, generator=g).to(DEV)
        gam, bet = torch.ones(C, device=DEV), torch.zeros(C, device=DEV)
        win = (torch.rand(2 * h, C, generator=g) - 0.5).to(DEV)
        wdw = (torch.rand(2 * h, 9, generator=g) - 0.5).to(DEV)
        wout = (torch.rand(C, h, generator=g) - 0.5).to(DEV)
        res = {}
        for mode in ("store", "nostore", "three"):
            r = Recorder(DEV, scratch_floats=1 << 20)
            Y, st = r.alloc(M, C), r.alloc(M, 2)
            T1, T2 = r.alloc(M, ldt)[:, :2 * hp], r.alloc(M, ldt)[:, :2 * hp]
            if mode == "three":
                r.gemm("in", x, win, T1, N=2 * hp, K=C, w_sn=C, w_sk=1, n_grp=(h, hp), pro=L.PRO_LN, ln=(gam, bet), stats=st)
                r.dwconv("dw", T1, wdw, T2, C_=2 * hp, dims=(B, S, S), grp=(h, hp))
                r.gemm("out", T2, wout, Y, N=C, K=h, w_sn=h, w_sk=1, pro=L.PRO_GATE, gate_off=hp, R=x)
            else:
                st_ = mode == "store"
                r.gdfn_fwd("g", x, (gam, bet), win, wdw, wout, Y, C_=C, h=h, hp=hp, dims=(B, S, S),
                           T1=T1 if st_ else None, T2=T2 if st_ else None, stats=st)
            res[mode] = timeit(r)
        print(f"C={C:3d} S={S:3d}: fused+store {res['store']*1e3:7.1f} us  fused {res['nostore']*1e3:7.1f} us  three kernels {res['three']*1e3:7.1f} us", flush=True)


if __name__ == "__main__":
    main()
