"""A/B helper: tools/bench_frfn.py with engine module constants overridden.  usage: python tools/ab/frfn_flag.py <net|ast> NAME=VALUE [...]"""
import ast
import os
import runpy
import sys
import torch
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root)
torch.cuda.set_device(0)
import underwater_image_restoration_amd.engine as e
what = sys.argv[1]
for kv in sys.argv[2:]:
    k, v = kv.split("=", 1)
    assert hasattr(e, k), k
    setattr(e, k, ast.literal_eval(v))
sys.argv = ["bench_frfn.py", "--what", what, "--no-cpu"]
runpy.run_path(os.path.join(root, "tools", "bench_frfn.py"), run_name="__main__")
