#!/bin/bash
O=gpurun_out/r04ar; mkdir -p $O
python -m pytest tests/test_hip_kernels.py -x -q -k "dwln" > $O/pytest.txt 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.txt
L=underwater_image_restoration_amd/libuir_hip.so
for i in 1 2; do
for v in old new; do
  cp tools/xp/libs/$v.so $L
  echo "== $v" | tee -a $O/fused.txt
  python tools/bench_fused.py 2>&1 | grep -i "dwln" | tee -a $O/fused.txt
done
done
for i in 1 2 3; do
for v in old new; do
  cp tools/xp/libs/$v.so $L
  python bench.py --no-cpu-baseline --no-kernel-breakdown --steps 10 --warmup 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', round(d['ms_per_step'],3), round(d['value'],1))" | tee -a $O/ab.txt
done
done
cp tools/xp/libs/new.so $L
