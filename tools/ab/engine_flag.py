"""A/B helper: bench.py's headline run with engine module constants overridden.  usage: python tools/ab/engine_flag.py NAME=VALUE [NAME=VALUE ...]"""
import ast
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
torch.cuda.set_device(0)
import bench
import underwater_image_restoration_amd.engine as e
for kv in sys.argv[1:]:
    k, v = kv.split("=", 1)
    assert hasattr(e, k), k
    setattr(e, k, ast.literal_eval(v))
sys.argv = ["bench.py", "--no-cpu-baseline", "--no-kernel-breakdown", "--steps", "10", "--warmup", "3"]
bench.main()
