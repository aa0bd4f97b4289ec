"""Copy one round-end evidence set from gpurun_out/<tag>/ into profiles/<round>/ under the z_ prefix (what DESIGN.md section 17 quotes).
usage: python tools/collect_evidence.py <tag> <round dir, e.g. r05>"""
import os
import shutil
import sys

tag, rd = sys.argv[1], sys.argv[2]
src, dst = os.path.join("gpurun_out", tag), os.path.join("profiles", rd)
os.makedirs(dst, exist_ok=True)
for f in ("bench.json", "kernel_stats.csv", "op_breakdown.csv", "stage_timeline.csv", "chain_gaps_events.txt", "bench_conv3b.txt", "bench_deep.txt", "bench_fused.txt", "bench_value_bwd.txt",
          "frfnnet_bench.json", "ast_bench.json", "smoke.log"):
    p = os.path.join(src, f)
    if os.path.exists(p):
        shutil.copy(p, os.path.join(dst, "z_" + f))
        print("copied", f)
