#!/bin/bash
# conv3b: parity tests + headline A/B (CONV3_SPLIT False / True) + untraced chain gaps
T=${1:-r05c}; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$T; mkdir -p $O
cd $R && python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1
timeout -k 10 600 python -m pytest tests/test_hip_kernels.py -x -q -m gpu -k "conv3x3" > $O/pytest_conv.txt 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_conv.txt
for i in 1 2 3; do
  python tools/ab/engine_flag.py CONV3_SPLIT=False 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | head -1 | sed 's/^/f32   /'
  python tools/ab/engine_flag.py CONV3_SPLIT=True 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | head -1 | sed 's/^/split /'
done | tee $O/ab.txt
timeout -k 10 300 python tools/bench_conv3b.py --tiles > $O/bench_conv3b.txt 2>&1 && echo conv3b-ok
timeout -k 10 300 python tools/chain_gaps_events.py 12 3 > $O/chain_gaps_events.txt 2>&1 && echo gaps-ok
timeout -k 10 900 python -m pytest tests/test_hip_model.py -x -q -m gpu > $O/pytest_model.txt 2>&1; echo "pytest model rc=$?"; tail -3 $O/pytest_model.txt
