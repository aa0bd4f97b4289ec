#!/bin/bash
O=gpurun_out/r04b; mkdir -p $O
python -m pytest tests/test_hip_kernels.py -m gpu -x -q -k "dwln or gdfn_fwd or attnv_fwd or qkv_fwd" > $O/pytest.txt 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.txt
python tools/bench_fused.py 20 > $O/fused_1.txt 2>&1; grep -v amdgpu $O/fused_1.txt
python tools/probes/stamps/run.py 2>&1 | grep -v amdgpu.ids > $O/stamps.txt; cat $O/stamps.txt
for m in 1 1; do python bench.py --no-cpu-baseline --no-kernel-breakdown --steps 10 --warmup 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench', d['ms_per_step'], d['value'])"; done
