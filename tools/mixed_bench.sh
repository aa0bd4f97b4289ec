#!/bin/bash
# Mixed-precision evidence (UIR_MIXED, knobs.py) on one MI355X, run through gpurun from the repo root:  tools/mixed_bench.sh <tag>
#   gpurun_out/<tag>/  wide-layer microbench f32 / bf16 / fp16, then the module-network train steps of BASELINE configs[3] / [4]
#   with the knob on.  These are their OWN metrics ("dtype": "bf16-mixed" / "fp16-mixed"); no f32 number is taken from here.
T=${1:-mixed}; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$T; mkdir -p $O
cd $R
timeout -k 10 200 python tools/microbench_wide.py 10 > $O/wide_f32.txt 2>&1 && echo wide-f32-ok
UIR_MIXED=bf16 timeout -k 10 200 python tools/microbench_wide.py 10 > $O/wide_bf16.txt 2>&1 && echo wide-bf16-ok
UIR_MIXED=fp16 timeout -k 10 200 python tools/microbench_wide.py 10 > $O/wide_fp16.txt 2>&1 && echo wide-fp16-ok
for K in bf16 fp16; do
  timeout -k 10 300 python tools/bench_frfn.py --what ast --no-cpu --mixed $K --breakdown-file $O/ast_${K}_breakdown.csv > $O/ast_${K}_bench.json 2> $O/ast_$K.err && echo ast-$K-ok
done
timeout -k 10 300 python tools/bench_frfn.py --what ast --no-cpu --mixed bf16 --loss fflMix > $O/ast_bf16_fflmix_bench.json 2> $O/ast_bf16_fflmix.err && echo ast-bf16-fflmix-ok
timeout -k 10 300 python tools/bench_frfn.py --what net --no-cpu --mixed bf16 --breakdown-file $O/frfnnet_bf16_breakdown.csv > $O/frfnnet_bf16_bench.json 2> $O/frfnnet_bf16.err && echo net-bf16-ok
# per-kernel times of the mixed AST step (rocprofv3 --kernel-trace --stats; the 16-bit contraction is uir_tiled_lp_kernel)
( cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_ast_bf16 -- python3 $R/tools/bench_frfn.py --what ast --steps 2 --warmup 1 --no-cpu --mixed bf16 > $O/prof_ast_bf16.log 2>&1 ) && echo prof-ast-bf16-ok
find $O -name "*kernel_trace.csv" -delete
