#!/bin/bash
O=gpurun_out/r04ae; mkdir -p $O
runh() { PYTHONPATH=. python3 tools/xp/ab_hp.py "$@" 2>$O/err.txt | grep '^{' | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('hp=$*', round(d['ms_per_step'],3), round(d['value'],1))" | tee -a $O/ab.txt; }
for i in 1 2 3; do
  runh 0
  runh 1
done
tail -3 $O/err.txt
