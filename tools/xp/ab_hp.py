"""A/B helper: the headline bench with the whole step issued on a high-priority stream (arg 1 = 1) or the default stream (0)."""
import sys
import torch
torch.cuda.set_device(0)
import bench
hp = sys.argv[1] == "1"
sys.argv = ["bench.py", "--no-cpu-baseline", "--no-kernel-breakdown", "--steps", "10", "--warmup", "3"]
if hp:
    s = torch.cuda.Stream(priority=-1)
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        bench.main()
else:
    bench.main()
