"""Calibrate FETCH_SIZE / WRITE_SIZE on known byte counts: a torch copy (wide 16 B/lane streaming) and this
repo's depthwise / GEMM kernels on tensors of known size.  Run under rocprofv3 --pmc (one counter per pass)."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from underwater_image_restoration_amd.engine import Recorder
DEV = "cuda:0"
M, C = 16 * 256 * 256, 192
x = torch.rand(M, C, device=DEV)                 # 805 MB
y = torch.empty_like(x)
for _ in range(2):
    y.copy_(x)                                    # known: read 805 MB, write 805 MB
torch.cuda.synchronize()
r = Recorder(DEV, scratch_floats=1 << 20)
w = torch.rand(C, 9, device=DEV)
Y = r.alloc(M, C)
r.dwconv("dw", x, w, Y, C_=C, dims=(16, 256, 256))          # known: read 805 MB (+ halo from L2), write 805 MB
x32 = torch.rand(M, 32, device=DEV)
W = torch.rand(96, 32, device=DEV)
Y2 = r.alloc(M, 96)
r.gemm("g", x32, W, Y2, N=96, K=32, w_sn=32, w_sk=1)        # known: read 134 MB, write 403 MB
r.run(r.fwd)
r.run(r.fwd)
torch.cuda.synchronize()
print("bytes: copy r/w 805306368; dwconv r/w 805306368; gemm r 134217728 w 402653184")
