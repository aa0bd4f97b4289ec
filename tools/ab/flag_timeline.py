"""Stage timeline of bench.py's headline run under engine constants: python tools/ab/flag_timeline.py <timeline csv> NAME=VALUE ..."""
import ast
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
torch.cuda.set_device(0)
import bench
import underwater_image_restoration_amd.engine as e
out = sys.argv[1]
for kv in sys.argv[2:]:
    k, v = kv.split("=", 1)
    assert hasattr(e, k), k
    setattr(e, k, ast.literal_eval(v))
sys.argv = ["bench.py", "--no-cpu-baseline", "--no-kernel-breakdown", "--steps", "10", "--warmup", "3", "--timeline-file", out]
bench.main()
