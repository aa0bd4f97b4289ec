#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out/r04ab; mkdir -p $O; R=$GRAFT_REPO_ROOT
python -m pytest tests/test_hip_model.py -x -q > $O/pytest_model.txt 2>&1; echo "pytest model rc=$?"; tail -2 $O/pytest_model.txt
runh() { PYTHONPATH=. python3 tools/xp/ab_engine_flag.py "$@" 2>$O/err.txt | grep '^{' | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('headline $*', round(d['ms_per_step'],3), round(d['value'],1))" | tee -a $O/ab.txt; }
for i in 1 2 3; do
  runh FAST_LANE_EVENTS=False
  runh FAST_LANE_EVENTS=True
done
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $O/trace -- python3 $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-kernel-breakdown > $O/trace.log 2>&1 && echo trace-ok
cd $R
python3 tools/trace_gaps.py $(find $O/trace -name "*kernel_trace.csv" | head -1) 3 | tee $O/gaps.txt | head -12
find $O -name "*kernel_trace.csv" -delete
