#!/bin/bash
O=gpurun_out/r04am; mkdir -p $O
python -m pytest tests/test_hip_model.py -x -q > $O/pytest_model.txt 2>&1; echo "pytest model rc=$?"; tail -2 $O/pytest_model.txt
runh() { PYTHONPATH=. python3 tools/xp/ab_engine_flag.py "$@" 2>$O/err.txt | grep '^{' | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('headline $*', round(d['ms_per_step'],3), round(d['value'],1))" | tee -a $O/ab.txt; }
for i in 1 2 3; do
  runh QKV16_LD=48
  runh QKV16_LD=64
done
