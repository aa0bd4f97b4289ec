import sys, torch
sys.path.insert(0, "/root/repo")
from underwater_image_restoration_amd.engine import Recorder
dev = "cuda"
def run(M, N, K, reps=20):
    r = Recorder(dev, scratch_floats=8 << 20)
    A, B, out = torch.randn(M, N, device=dev), torch.randn(M, K, device=dev), r.alloc(N, K)
    r.gram("g", A, B, out, N=N, K=K, o_sn=K, o_sk=1, own_slab=True)
    for _ in range(3): r.run(r.fwd)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): r.run(r.fwd)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / reps * 1e3
    print(f"gram M={M:8d} N={N:4d} K={K:4d}: {us:8.1f} us  {4 * M * (N + K) / us / 1e6:5.2f} TB/s  {2 * M * N * K / us / 1e6:6.1f} TFLOP/s  ({len(r.fwd)} launches)", flush=True)
for M in (1048576, 262144, 65536, 16384):
    for N, K in ((32, 32), (64, 64), (64, 32), (128, 32), (96, 32), (128, 64)):
        run(M, N, K)
