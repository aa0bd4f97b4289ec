#!/bin/bash
O=gpurun_out/r04ac; mkdir -p $O
python -m pytest tests/test_hip_kernels.py -x -q -k "gram or conv3x3" > $O/pytest.txt 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.txt
L=underwater_image_restoration_amd/libuir_hip.so
for v in old new; do
  cp tools/xp/libs/$v.so $L
  echo "== $v" | tee -a $O/conv3.txt
  python tools/bench_conv3.py 2>&1 | grep "dW\|red" | tee -a $O/conv3.txt
done
for i in 1 2 3; do
for v in old new; do
  cp tools/xp/libs/$v.so $L
  python bench.py --no-cpu-baseline --no-kernel-breakdown --steps 10 --warmup 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', round(d['ms_per_step'],3), round(d['value'],1))" | tee -a $O/ab.txt
done
done
cp tools/xp/libs/new.so $L
python -m pytest tests/test_hip_model.py -x -q 2>&1 | tail -2
