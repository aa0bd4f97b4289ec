#!/bin/bash
# A/B of a SOURCE edit on one box: tools/ab/src_ab.sh <tag> <file> <sed expression that turns the feature OFF> [pytest -k expression]
# builds the tree as it is (ON), runs the kernel tests + bench_deep + 3 bench lines, applies the sed to the box's copy, rebuilds, 3 bench lines (OFF), interleaved twice
T=$1; F=$2; SED=$3; K=${4:-ts_gemm}; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$T; mkdir -p $O
cd $R && python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1
timeout -k 10 900 python -m pytest tests/test_hip_kernels.py tests/test_tiled.py -x -q -m gpu -k "$K" > $O/pytest.txt 2>&1; rc=$?; echo "pytest rc=$rc"; tail -5 $O/pytest.txt
[ $rc -eq 0 ] || exit 1
python tools/bench_deep.py 30 > $O/bench_deep_on.txt 2>&1
cp $F $O/on.src
sed "$SED" $F > $O/off.src
cmp -s $O/on.src $O/off.src && { echo "sed changed nothing"; exit 1; }
for i in 1 2; do
  cp $O/on.src $F && python -c "import __graft_entry__ as g; g.build()" >> $O/build.log 2>&1
  for j in 1 2; do python bench.py --no-cpu-baseline 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | head -1 | sed 's/^/on  /'; done
  cp $O/off.src $F && python -c "import __graft_entry__ as g; g.build()" >> $O/build.log 2>&1
  [ $i -eq 1 ] && python tools/bench_deep.py 30 > $O/bench_deep_off.txt 2>&1
  for j in 1 2; do python bench.py --no-cpu-baseline 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | head -1 | sed 's/^/off /'; done
done | tee $O/ab.txt
cp $O/on.src $F
paste -d'|' <(grep -v amdgpu $O/bench_deep_on.txt | cut -c1-72) <(grep -v amdgpu $O/bench_deep_off.txt | cut -c46-72) | grep "bwd\|ups\|per-sample"
