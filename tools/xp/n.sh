#!/bin/bash
O=gpurun_out/r04n; mkdir -p $O
python -m pytest tests/test_hip_kernels.py -m gpu -x -q -k "dft or spectral" > $O/pytest.txt 2>&1; echo "pytest rc=$?"; tail -6 $O/pytest.txt
python -m pytest tests/test_hip_model.py -m gpu -x -q > $O/pytest_model.txt 2>&1; echo "pytest model rc=$?"; tail -2 $O/pytest_model.txt
b() { python bench.py --no-cpu-baseline --no-kernel-breakdown --steps 10 --warmup 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['ms_per_step'],3), round(d['value'],1))"; }
UIR_XP_FDFT="" b unfused; b fused-all; UIR_XP_FDFT="32,64" b fused-32-64; UIR_XP_FDFT="" b unfused; b fused-all; UIR_XP_FDFT="32,64" b fused-32-64
