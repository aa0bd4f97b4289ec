"""What this box's HBM gives plain streaming kernels (torch's own, float4 per lane): pure write (fill_), pure read (sum), copy, add -
the yardstick for the write-heavy contractions (a 64 -> 256 Linear on 1 M tokens writes 4x what it reads)."""
import time
import torch
n = 1 << 28                      # 1 GiB of f32
x = torch.empty(n, device="cuda"); y = torch.empty(n, device="cuda"); z = torch.empty(n, device="cuda")
x.normal_(); y.normal_()
def t(f, nbytes, name, reps=20):
    f(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): f()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    print(f"{name:28s} {dt * 1e6:8.1f} us  {nbytes / dt / 1e12:5.2f} TB/s", flush=True)
t(lambda: z.fill_(1.0), 4 * n, "fill (write only)")
t(lambda: x.sum(), 4 * n, "sum (read only)")
t(lambda: z.copy_(x), 8 * n, "copy (1 read + 1 write)")
t(lambda: torch.add(x, y, out=z), 12 * n, "add (2 reads + 1 write)")
