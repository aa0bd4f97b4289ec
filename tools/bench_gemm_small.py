import sys, torch
sys.path.insert(0, "/root/repo")
from underwater_image_restoration_amd.engine import Recorder
dev = "cuda"
def run(M, K, N, bias=True, reps=30):
    r = Recorder(dev, scratch_floats=1 << 20)
    X, W, Y = torch.randn(M, K, device=dev), torch.randn(N, K, device=dev), r.alloc(M, N)
    b = torch.randn(N, device=dev) if bias else None
    r.gemm("g", X, W, Y, N=N, K=K, w_sn=K, w_sk=1, bias=b)
    for _ in range(3): r.run(r.fwd)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): r.run(r.fwd)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / reps * 1e3
    gb = 4 * M * (K + N) / 1e9
    print(f"M={M:8d} K={K:4d} N={N:4d}: {us:8.1f} us  {gb / us * 1e6 / 1e3:6.2f} TB/s  {2 * M * K * N / us / 1e6:6.1f} TFLOP/s", flush=True)
for M in (524288, 131072, 32768):
    for K, N in ((32, 32), (32, 64), (64, 32), (64, 64), (32, 96), (128, 128), (64, 128), (128, 64)):
        run(M, K, N)
