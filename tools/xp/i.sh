#!/bin/bash
O=gpurun_out/r04i; mkdir -p $O
UIR_FUSED_SPLIT=1 python tools/bench_deep.py 30 bwd > $O/deep_1b.txt 2>&1; grep -v "amdgpu\|\[uir\]" $O/deep_1b.txt | grep "dt1\|dt2 one\|dqk\|dgate"
python -m pytest tests/test_hip_kernels.py -m gpu -x -q -k "tall_skinny" 2>&1 | tail -2
b() { python bench.py --no-cpu-baseline --no-kernel-breakdown --steps 10 --warmup 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['ms_per_step'],3), round(d['value'],1))"; }
b ts; b ts
