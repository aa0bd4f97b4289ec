#!/bin/bash
O=gpurun_out/r04c; mkdir -p $O
python -m pytest tests/test_hip_model.py -m gpu -x -q > $O/pytest.txt 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.txt
b() { python bench.py --no-cpu-baseline --no-kernel-breakdown --steps 10 --warmup 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['ms_per_step'],3), round(d['value'],1))"; }
UIR_XP_INRED=999 UIR_XP_SIDE=999 b "off        "
UIR_XP_INRED=64 UIR_XP_SIDE=999 b "inred64    "
UIR_XP_INRED=999 UIR_XP_SIDE=64 b "side64     "
UIR_XP_INRED=64 UIR_XP_SIDE=64 b "both64     "
UIR_XP_INRED=32 UIR_XP_SIDE=64 b "inred32s64 "
UIR_XP_INRED=64 UIR_XP_SIDE=32 b "inred64s32 "
UIR_XP_INRED=999 UIR_XP_SIDE=999 b "off        "
UIR_XP_INRED=64 UIR_XP_SIDE=64 b "both64     "
