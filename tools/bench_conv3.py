"""Microbench of the 3x3 convolutions with <= 32 channels on both sides (forward and input gradient) at the headline's sizes.
usage: python tools/bench_conv3.py   (one line per shape: us per launch, algorithmic GB/s)"""
import ctypes as C
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from underwater_image_restoration_amd import engine as E   # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    shapes = [("outputl", 16, 256, 32, 8), ("output", 16, 256, 8, 4), ("down1", 16, 256, 16, 8), ("down2", 16, 128, 32, 16)]
    for name, B, S, Cin, Cout in shapes:
        M = B * S * S
        r = E.Recorder(dev)
        x, w = torch.randn(M, Cin, device=dev), torch.randn(Cout, Cin, 3, 3, device=dev)
        y, dy, dx, res = r.alloc(M, Cout), torch.randn(M, Cout, device=dev), r.alloc(M, Cin), torch.randn(M, Cin, device=dev)
        ng = dict(n_grp=(3, 4)) if name == "output" else {}
        kg = dict(k_grp=(3, 4)) if name == "output" else {}
        if name == "output":
            w = torch.randn(3, Cin, 3, 3, device=dev)
        r.gemm(name, x, w, y, N=Cout, K=Cin, w_sn=Cin * 9, w_sk=9, w_st=1, taps=9, hw=(S, S), **ng)
        r.gemm(name + ".dX", dy, w, dx, N=Cin, K=Cout, w_sn=9, w_sk=Cin * 9, w_st=1, taps=9, flip=1, hw=(S, S), **kg)
        r.gemm(name + ".dX+R", dy, w, dx, N=Cin, K=Cout, w_sn=9, w_sk=Cin * 9, w_st=1, taps=9, flip=1, hw=(S, S), R=res, **kg)
        dw = r.alloc(w.shape[0], Cin, 3, 3)
        r.gram(name + ".dW", dy, x, dw, N=Cout, K=Cin, o_sn=Cin * 9, o_sk=9, o_st=1, taps=9, hw=(S, S), **ng)
        sp = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        for i, (fn, args, label) in enumerate(r.fwd):
            for _ in range(3):
                assert fn(*args, sp) == 0, label
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                fn(*args, sp)
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1000 / 20
            if fn is None:
                continue
            nb = 4 * M * (Cin + Cout + (Cin if i == 2 else 0))
            print(f"{label:14s} {us:8.1f} us  {nb / us / 1e3:7.0f} GB/s", flush=True)


if __name__ == "__main__":
    main()
