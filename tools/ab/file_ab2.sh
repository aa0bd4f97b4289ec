#!/bin/bash
# A/B of one source FILE on one box with any microbench: tools/ab/file_ab2.sh <tag> <file in the tree> <alternative version> <pytest -k expression> "<microbench command>"
# builds the tree as it is (NEW), kernel tests, the microbench; then alternates NEW / OLD builds with two bench lines each, twice (the microbench once on OLD)
T=$1; F=$2; ALT=$3; K=$4; MB=$5; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$T; mkdir -p $O
cd $R && python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1
timeout -k 10 900 python -m pytest tests/test_hip_kernels.py -x -q -m gpu -k "$K" > $O/pytest.txt 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 $O/pytest.txt
[ $rc -eq 0 ] || exit 1
$MB > $O/mb_new.txt 2>&1
cp $F $O/new.src
for i in 1 2; do
  cp $O/new.src $F && python -c "import __graft_entry__ as g; g.build()" >> $O/build.log 2>&1
  for j in 1 2; do python bench.py --no-cpu-baseline --no-kernel-breakdown --steps 10 --warmup 3 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | head -1 | sed 's/^/new /'; done
  cp $ALT $F && python -c "import __graft_entry__ as g; g.build()" >> $O/build.log 2>&1
  [ $i -eq 1 ] && $MB > $O/mb_old.txt 2>&1
  for j in 1 2; do python bench.py --no-cpu-baseline --no-kernel-breakdown --steps 10 --warmup 3 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | head -1 | sed 's/^/old /'; done
done | tee $O/ab.txt
cp $O/new.src $F
echo "--- microbench NEW | OLD"; paste -d'|' <(grep -v "amdgpu" $O/mb_new.txt | cut -c1-90) <(grep -v "amdgpu" $O/mb_old.txt | cut -c1-90)
