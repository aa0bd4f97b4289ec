#!/bin/bash
T=${1:-r05_full}; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$T; mkdir -p $O
cd $R && python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1
timeout -k 10 1100 python -m pytest tests/ -x -q -m gpu > $O/full_gpu_pytest.txt 2>&1; echo "pytest rc=$?"; tail -5 $O/full_gpu_pytest.txt
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1 && echo smoke-ok; tail -3 $O/smoke.log
