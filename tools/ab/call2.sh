#!/bin/bash
# conv3b: parity tests + microbench + headline A/B (CONV3_SPLIT False / True), then the diagnostic skip arms
T=${1:-r05b}; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$T; mkdir -p $O
cd $R
timeout -k 10 600 python -m pytest tests/test_hip_kernels.py -x -q -m gpu -k "conv3x3" > $O/pytest_conv.txt 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest_conv.txt
timeout -k 10 300 python tools/bench_conv3b.py > $O/bench_conv3b.txt 2>&1 && echo conv3b-ok; cat $O/bench_conv3b.txt
for i in 1 2; do
  python tools/ab/engine_flag.py CONV3_SPLIT=False 2>&1 | grep -o '"ms_per_step": [0-9.]*' | head -1 | sed 's/^/f32   /'
  python tools/ab/engine_flag.py CONV3_SPLIT=True 2>&1 | grep -o '"ms_per_step": [0-9.]*' | head -1 | sed 's/^/split /'
done | tee $O/ab.txt
