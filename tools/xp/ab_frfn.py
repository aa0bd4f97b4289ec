"""A/B helper: tools/bench_frfn.py with engine module constants overridden.  usage: ab_frfn.py net|ast NAME=VALUE ..."""
import ast
import runpy
import sys
import torch
torch.cuda.set_device(0)
import underwater_image_restoration_amd.engine as e
import underwater_image_restoration_amd.model.block as blk
what = sys.argv[1]
for kv in sys.argv[2:]:
    k, v = kv.split("=", 1)
    tgt = blk if k.startswith("block.") else e
    k = k.split(".")[-1]
    assert hasattr(tgt, k), k
    setattr(tgt, k, ast.literal_eval(v))
sys.argv = ["bench_frfn.py", "--what", what, "--no-cpu"]
runpy.run_path("tools/bench_frfn.py", run_name="__main__")
