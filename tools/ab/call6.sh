#!/bin/bash
# gramb: parity tests, deep-level microbench with the old (UIR_FUSED_SPLIT=0 routes Grams to csrc/gram.hip) and new paths, headline A/B
T=${1:-r05f}; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$T; mkdir -p $O
cd $R && python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1
timeout -k 10 900 python -m pytest tests/test_hip_kernels.py -x -q -m gpu -k "gram or conv3x3" > $O/pytest_gram.txt 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_gram.txt
UIR_FUSED_SPLIT=0 timeout -k 10 300 python tools/bench_deep.py 30 "Gram" > $O/bench_deep_old.txt 2>&1
UIR_FUSED_SPLIT=0 timeout -k 10 300 python tools/bench_deep.py 30 "weight gradient" >> $O/bench_deep_old.txt 2>&1
timeout -k 10 300 python tools/bench_deep.py 30 "Gram" > $O/bench_deep_new.txt 2>&1
timeout -k 10 300 python tools/bench_deep.py 30 "weight gradient" >> $O/bench_deep_new.txt 2>&1
paste -d'\n' $O/bench_deep_old.txt $O/bench_deep_new.txt | grep -v amdgpu
for i in 1 2 3; do
  python tools/ab/engine_flag.py GRAM_SPLIT=False 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | head -1 | sed 's/^/f32 grams   /'
  python tools/ab/engine_flag.py GRAM_SPLIT=True 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | head -1 | sed 's/^/split grams /'
done | tee $O/ab.txt
timeout -k 10 900 python -m pytest tests/test_hip_model.py -x -q -m gpu > $O/pytest_model.txt 2>&1; echo "pytest model rc=$?"; tail -3 $O/pytest_model.txt
