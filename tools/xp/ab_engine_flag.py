"""A/B helper: bench.py's headline run with one engine module constant overridden.  usage: ab_engine_flag.py NAME VALUE [bench args]"""
import ast
import sys
import torch
torch.cuda.set_device(0)
import bench
import underwater_image_restoration_amd.engine as e
setattr(e, sys.argv[1], ast.literal_eval(sys.argv[2]))
sys.argv = ["bench.py", "--no-cpu-baseline", "--no-kernel-breakdown", "--steps", "10", "--warmup", "3"] + sys.argv[3:]
bench.main()
