#!/bin/bash
O=gpurun_out/r04e; mkdir -p $O
python -m pytest tests/test_hip_kernels.py -m gpu -x -q -k "dwln or gdfn_fwd or gate" > $O/pytest.txt 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest.txt
python -m pytest tests/test_hip_model.py -m gpu -x -q > $O/pytest_model.txt 2>&1; echo "pytest model rc=$?"; tail -3 $O/pytest_model.txt
b() { python bench.py --no-cpu-baseline --no-kernel-breakdown --steps 10 --warmup 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['ms_per_step'],3), round(d['value'],1))"; }
b "il"; b "il"
python bench.py --no-cpu-baseline --steps 10 --warmup 3 --breakdown-file $O/op_breakdown.csv --timeline-file $O/timeline.csv > $O/bench.json 2>$O/bench.err
head -12 $O/op_breakdown.csv
