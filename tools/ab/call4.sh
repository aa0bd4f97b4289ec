#!/bin/bash
# conv3b tile defaults + P launches at the start: parity tests, then A/B of the stage hold of deep-level bulk launches
T=${1:-r05d}; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$T; mkdir -p $O
cd $R && python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1
timeout -k 10 600 python -m pytest tests/test_hip_kernels.py -x -q -m gpu -k "conv3x3" > $O/pytest_conv.txt 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_conv.txt
for i in 1 2 3; do
  python tools/ab/engine_flag.py 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | head -1 | sed 's/^/base      /'
  python tools/ab/engine_flag.py 'BULK_HOLD_STAGES=("bwd.ups_2","bwd.enc2(L2)")' 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | head -1 | sed 's/^/hold deep /'
  python tools/ab/engine_flag.py 'BULK_HOLD_STAGES=("bwd.dec0(L2)","bwd.enc2(L2)")' 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | head -1 | sed 's/^/hold L3   /'
done | tee $O/ab.txt
timeout -k 10 900 python -m pytest tests/test_hip_model.py -x -q -m gpu > $O/pytest_model.txt 2>&1; echo "pytest model rc=$?"; tail -3 $O/pytest_model.txt
