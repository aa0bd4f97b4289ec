#!/bin/bash
O=gpurun_out/r04d; mkdir -p $O
python -m pytest tests/test_hip_kernels.py -m gpu -x -q -k "dwln or gdfn_fwd" > $O/pytest.txt 2>&1; echo "pytest rc=$?"; tail -2 $O/pytest.txt
UIR_FUSED_SPLIT=2 python -m pytest tests/test_hip_kernels.py -m gpu -x -q -k "gdfn_fwd_fused" > $O/pytest2.txt 2>&1; echo "pytest(spread) rc=$?"; tail -2 $O/pytest2.txt
for m in 1 2; do UIR_FUSED_SPLIT=$m python tools/bench_fused.py 20 > $O/fused_$m.txt 2>&1; done
grep -h "gdfn_fwd" $O/fused_1.txt $O/fused_2.txt; grep -h dwln $O/fused_1.txt
b() { python bench.py --no-cpu-baseline --no-kernel-breakdown --steps 10 --warmup 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['ms_per_step'],3), round(d['value'],1))"; }
UIR_FUSED_SPLIT=1 b "split=1"; UIR_FUSED_SPLIT=2 b "split=2(spread)"; UIR_FUSED_SPLIT=1 b "split=1"; UIR_FUSED_SPLIT=2 b "split=2(spread)"
python tools/probes/stamps/run.py 2>&1 | grep -v amdgpu.ids > $O/stamps.txt; grep -A8 "gdfn layout" $O/stamps.txt | grep "all\|product"
