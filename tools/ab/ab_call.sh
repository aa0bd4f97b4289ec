#!/bin/bash
# one GPU call: build, the kernel tests named by $2, a microbench, an A/B of one engine flag ($3, e.g. FUSE_VALUE_BWD_64), the model tests
T=${1:-r05r}; K=${2:-attnv_bwd}; F=${3:-FUSE_VALUE_BWD_64}; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$T; mkdir -p $O
cd $R && python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1
timeout -k 10 600 python -m pytest tests/test_hip_kernels.py -x -q -m gpu -k "$K" > $O/pytest.txt 2>&1; rc=$?; echo "pytest rc=$rc"; tail -12 $O/pytest.txt
[ $rc -eq 0 ] || exit 1
timeout -k 10 300 python tools/bench_value_bwd.py > $O/bench_value_bwd.txt 2>&1 && grep -v amdgpu $O/bench_value_bwd.txt && \
for i in 1 2 3; do
  python tools/ab/engine_flag.py $F=False 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | head -1 | sed 's/^/off /'
  python tools/ab/engine_flag.py $F=True 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | head -1 | sed 's/^/on  /'
done | tee $O/ab.txt
timeout -k 10 900 python -m pytest tests/test_hip_model.py -x -q -m gpu > $O/pytest_model.txt 2>&1; echo "pytest model rc=$?"; tail -3 $O/pytest_model.txt
