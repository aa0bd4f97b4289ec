#!/bin/bash
O=gpurun_out/r04aq; mkdir -p $O
UIR_FUSED_SPLIT=0 python -m pytest tests/test_hip_model.py tests/test_hip_kernels.py -x -q > $O/pytest_f32.txt 2>&1; echo "f32-mfma rc=$?"; tail -2 $O/pytest_f32.txt
UIR_FUSED_SPLIT=0 python bench.py --no-cpu-baseline --no-kernel-breakdown --steps 5 --warmup 2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('UIR_FUSED_SPLIT=0', round(d['ms_per_step'],3), round(d['value'],1))"
