"""Accuracy of the split-bf16 wide contraction (UIR_TILED_SPLIT=1: csrc/tiled.hip, uir_tiled_split_kernel) against fp64, next to the
f32-MFMA kernel (run without the variable) on the same inputs.  Prints max |err| / max |ref| and the rms relative error."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from underwater_image_restoration_amd.engine import Recorder
dev = "cuda"
g = torch.Generator().manual_seed(0)
for M, K, N in ((4096, 2048, 512), (16384, 512, 2048), (65536, 256, 1024)):
    x, w, dy = (torch.rand(M, K, generator=g, dtype=torch.float64) * 2 - 1), (torch.rand(N, K, generator=g, dtype=torch.float64) * 2 - 1) * K ** -0.5, \
        (torch.rand(M, N, generator=g, dtype=torch.float64) * 2 - 1)
    # wide dynamic range: scale rows by powers of two
    x = x * torch.exp2(torch.randint(-6, 7, (M, 1), generator=g).double())
    r = Recorder(dev, scratch_floats=16 << 20)
    X, W, DY = x.float().to(dev), w.float().to(dev), dy.float().to(dev)
    Y, DX, DW = r.alloc(M, N), r.alloc(M, K), r.alloc(N, K)
    r.gemm("f", X, W, Y, N=N, K=K, w_sn=K, w_sk=1)
    r.gemm("dx", DY, W, DX, N=K, K=N, w_sn=1, w_sk=K)
    r.gram("dw", DY, X, DW, N=N, K=K, o_sn=K, o_sk=1, own_slab=True)
    r.run(r.fwd); torch.cuda.synchronize()
    xd, wd, dyd = X.double().cpu(), W.double().cpu(), DY.double().cpu()
    for name, got, ref in (("fwd", Y, xd @ wd.T), ("dX ", DX, dyd @ wd), ("dW ", DW, dyd.T @ xd)):
        e = (got.double().cpu() - ref)
        print(f"M={M} K={K} N={N} {name}: max|err|/max|ref| {float(e.abs().max() / ref.abs().max()):.2e}   rms err / rms ref {float(e.pow(2).mean().sqrt() / ref.pow(2).mean().sqrt()):.2e}", flush=True)
