#!/bin/bash
O=gpurun_out/r04g; mkdir -p $O
python -m pytest tests/test_hip_kernels.py -m gpu -x -q -k "dwln" > $O/pytest.txt 2>&1; echo "pytest rc=$?"; tail -2 $O/pytest.txt
UIR_FUSED_SPLIT=0 python -m pytest tests/test_hip_kernels.py -m gpu -x -q -k "dwln" > $O/pytest0.txt 2>&1; echo "pytest(f32) rc=$?"; tail -2 $O/pytest0.txt
for m in 0 1; do echo "UIR_FUSED_SPLIT=$m"; UIR_FUSED_SPLIT=$m python tools/bench_fused.py 20 dwln 2>&1 | grep dwln; done
b() { python bench.py --no-cpu-baseline --no-kernel-breakdown --steps 10 --warmup 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['ms_per_step'],3), round(d['value'],1))"; }
UIR_FUSED_SPLIT=0 b f32; UIR_FUSED_SPLIT=1 b split; UIR_FUSED_SPLIT=0 b f32; UIR_FUSED_SPLIT=1 b split
