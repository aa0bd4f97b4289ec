#!/bin/bash
# experiment set a (round 4): split-bf16 inside the tile-fused forward kernels
O=gpurun_out/r04a; mkdir -p $O
python -m pytest tests/test_hip_kernels.py -m gpu -x -q -k "gdfn_fwd or attnv_fwd or qkv_fwd" > $O/pytest.txt 2>&1; echo "pytest rc=$?"
tail -3 $O/pytest.txt
for m in 0 1; do UIR_FUSED_SPLIT=$m python tools/bench_fused.py 20 > $O/fused_$m.txt 2>&1; done
paste -d'|' $O/fused_0.txt $O/fused_1.txt | cut -c1-140
for m in 0 1 0 1; do UIR_FUSED_SPLIT=$m python bench.py --no-cpu-baseline --no-kernel-breakdown --steps 10 --warmup 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('split=$m', d['ms_per_step'], d['value'])"; done
UIR_FUSED_SPLIT=1 python bench.py --no-cpu-baseline --steps 10 --warmup 3 --breakdown-file $O/op_breakdown_1.csv --timeline-file $O/timeline_1.csv > $O/bench_1.json 2>$O/bench_1.err
