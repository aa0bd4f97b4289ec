"""Microbench of UpSample1's 3x3 convolutions (SpectralTransformer.py:196; C = 128 / 64 / 32 at the headline's sizes): forward, input
gradient and weight gradient on the nine-tap f32 MFMA paths of uir_gemm / uir_gram and on the split-bf16 kernels of csrc/conv3b.hip.
usage: python tools/bench_conv3b.py [--tiles]   (--tiles: also every forced workgroup tile; one line per launch: us, TFLOP/s f32-equivalent)"""
import ctypes as C
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from underwater_image_restoration_amd import engine as E   # noqa: E402


def main():
    tiles = "--tiles" in sys.argv
    dev = torch.device("cuda", 0)
    shapes = [("ups_1", 16, 32, 128, 256), ("ups_2", 16, 64, 64, 128), ("ups_3", 16, 128, 32, 64), ("down3", 16, 64, 64, 32)]
    for name, B, S, Cin, Cout in shapes:
        M = B * S * S
        dims = (B, S, S)
        r = E.Recorder(dev)
        x, w = torch.randn(M, Cin, device=dev), torch.randn(Cout, Cin, 3, 3, device=dev) * (Cin * 9) ** -0.5
        y, dy, dx = r.alloc(M, Cout), torch.randn(M, Cout, device=dev), r.alloc(M, Cin)
        dw = r.alloc(Cout, Cin, 3, 3)
        r.gemm(name + " fwd  f32", x, w, y, N=Cout, K=Cin, w_sn=Cin * 9, w_sk=9, w_st=1, taps=9, hw=(S, S))
        r.gemm(name + " dX   f32", dy, w, dx, N=Cin, K=Cout, w_sn=9, w_sk=Cin * 9, w_st=1, taps=9, flip=1, hw=(S, S))
        r.gram(name + " dW   f32", dy, x, dw, N=Cout, K=Cin, o_sn=Cin * 9, o_sk=9, o_st=1, taps=9, hw=(S, S))
        pf, pb = r.conv3_wsplit(name + " wsplit", w, Cout, Cin, want_bwd=True)
        y2, dx2, dw2 = r.alloc(M, Cout), r.alloc(M, Cin), r.alloc(Cout, Cin, 3, 3)
        r.conv3_split(name + " fwd  split", x, pf, y2, N=Cout, K=Cin, dims=dims)
        r.conv3_split(name + " dX   split", dy, pb, dx2, N=Cin, K=Cout, dims=dims)
        if tiles:       # every tile the shape admits, forced (UirConv3bArgs.tile)
            for tm, tn in ((256, 32), (128, 32), (128, 128), (64, 128), (128, 64), (64, 64)):
                for tag, (xx, pp, yy, nn, kk) in (("fwd", (x, pf, y2, Cout, Cin)), ("dX ", (dy, pb, dx2, Cin, Cout))):
                    if nn % tn == 0 and S % (tm // 16) == 0 and (tn == 32) == (nn % 64 != 0 or tn == 32) and not (tn == 32 and nn % 64 == 0 and nn > 64):
                        r.conv3_split(f"{name} {tag}  {tm}x{tn}", xx, pp, yy, N=nn, K=kk, dims=dims, tile=tm * 1000 + tn)
        if E.Recorder.conv3_split_wgrad_ok(Cout, Cin, dims):
            r.conv3_split_wgrad(name + " dW   split", dy, x, dw2, N=Cout, K=Cin, dims=dims)
        sp = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        for fn, args, label in r.fwd:
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
            fl = 2.0 * M * Cin * Cout * 9
            tf = f"{fl / us / 1e6:7.1f} TFLOP/s" if ("red" not in label and "wsplit" not in label) else ""
            print(f"{label:20s} {us:8.1f} us  {tf}", flush=True)
        torch.cuda.synchronize()
        e = lambda a, b: float((a - b).abs().max() / b.abs().max())
        print(f"   split vs f32: fwd {e(y2, y):.2e}  dX {e(dx2, dx):.2e}  dW {e(dw2, dw):.2e}", flush=True)


if __name__ == "__main__":
    main()
