"""Wide Linear shapes of AST / NewBigFRFNModel in isolation: forward (X W^T), input gradient (dY W), weight gradient (dY^T X).
Run twice to compare the dispatch:  UIR_TILED_MINK=0 UIR_TILED_GRAM_MIN=0 python tools/microbench_wide.py   (previous kernels)
                                    python tools/microbench_wide.py                                          (csrc/tiled.hip)"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from underwater_image_restoration_amd.engine import Recorder
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
rnd = lambda *s: torch.rand(*s, device=dev, generator=g) * 2 - 1
# (tokens M, in C, out N): AST 512^2 bs 4 levels (dim, 4 dim) and the attention projections
SHAPES = [(1048576, 64, 256), (262144, 128, 512), (65536, 256, 1024), (16384, 512, 2048), (4096, 512, 2048), (16384, 512, 1024), (65536, 256, 256), (262144, 64, 256)]
for M, Cin, N in SHAPES:
    r = Recorder(dev, scratch_floats=16 << 20)
    x, w, dy = rnd(M, Cin), rnd(N, Cin) * .05, rnd(M, N)
    y, dx, dw = r.alloc(M, N), r.alloc(M, Cin), r.alloc(N, Cin)
    cases = {}
    def case(name, fn):
        r.fwd = []; r.cur = r.fwd; r.meta[id(r.fwd)] = []; r.lanes[id(r.fwd)] = []; r.variants[id(r.fwd)] = []
        fn(); cases[name] = r.fwd
    case("fwd ", lambda: r.gemm("f", x, w, y, N=N, K=Cin, w_sn=Cin, w_sk=1))
    case("fwd2", lambda: r.gemm("f2", dy, w, dx, N=Cin, K=N, w_sn=1, w_sk=Cin))          # K = wide side (linear2 forward = same shape as dX)
    case("dW  ", lambda: r.gram("w", dy, x, dw, N=N, K=Cin, o_sn=Cin, o_sk=1, own_slab=True))
    fl = 2.0 * M * Cin * N
    for name, ops in cases.items():
        r.run(ops); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps): r.run(ops)
        torch.cuda.synchronize()
        us = (time.perf_counter() - t0) / reps * 1e6
        print(f"M={M:8d} {Cin:5d}->{N:5d} {name} {us:9.1f} us {fl / us / 1e6:7.1f} TFLOP/s", flush=True)
