#!/bin/bash
T=${1:-r05p}; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$T; mkdir -p $O
cd $R && python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1
for i in 1 2 3; do
  python tools/ab/engine_flag.py GRAM_SLAB_CAP=False 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | head -1 | sed 's/^/no cap /'
  python tools/ab/engine_flag.py GRAM_SLAB_CAP=True 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | head -1 | sed 's/^/cap    /'
done | tee $O/ab.txt
timeout -k 10 600 python -m pytest tests/test_hip_model.py -x -q -m gpu -k "oracle or golden or fixture" > $O/pytest_model.txt 2>&1; echo "pytest model rc=$?"; tail -2 $O/pytest_model.txt
