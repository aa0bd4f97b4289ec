#!/bin/bash
# one GPU call: build, kernel tests named by $2, an A/B of one engine flag ($3: NAME, arms False / True), the model tests named by $4
T=${1:-r05k}; K=${2:-attnv}; F=${3:-RECOMPUTE_U}; MK=${4:-"full_model or trajectory_cfg1_vs or full_size or transformer_block"}; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$T; mkdir -p $O
cd $R && python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1
timeout -k 10 600 python -m pytest tests/test_hip_kernels.py -x -q -m gpu -k "$K" > $O/pytest.txt 2>&1; rc=$?; echo "pytest rc=$rc"; tail -5 $O/pytest.txt
[ $rc -eq 0 ] || exit 1
bash tools/ab/flags_multi.sh $T 3 "$F=False" "$F=True"
timeout -k 10 900 python -m pytest tests/test_hip_model.py -x -q -m gpu -k "$MK" > $O/pytest_model.txt 2>&1; echo "pytest model rc=$?"; tail -3 $O/pytest_model.txt
