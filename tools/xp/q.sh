#!/bin/bash
O=gpurun_out/r04y2; mkdir -p $O
run() { PYTHONPATH=. python3 tools/xp/ab_frfn.py "$@" 2>$O/err.txt | grep '^{' | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$*', round(d['ms_per_step'],3), round(d['value'],1))" | tee -a $O/ab.txt; }
runh() { PYTHONPATH=. python3 tools/xp/ab_engine_flag.py "$@" 2>$O/err.txt | grep '^{' | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('headline $*', round(d['ms_per_step'],3), round(d['value'],1))" | tee -a $O/ab.txt; }
for i in 1 2; do
  run net BULK_DEFER=0
  run net BULK_DEFER=30
  run ast BULK_DEFER=0
  run ast BULK_DEFER=30
  runh BULK_DEFER=0
  runh BULK_DEFER=30
done
python -m pytest tests/test_hip_model.py -x -q 2>&1 | tail -2
