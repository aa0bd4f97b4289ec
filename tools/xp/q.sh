#!/bin/bash
O=gpurun_out/r04ap; mkdir -p $O
python -m pytest tests/test_hip_kernels.py -x -q -k "attnv or fused" > $O/pytest.txt 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.txt
L=underwater_image_restoration_amd/libuir_hip.so
for i in 1 2; do
for v in old new; do
  cp tools/xp/libs/$v.so $L
  echo "== $v" | tee -a $O/fused.txt
  python tools/bench_fused.py 2>&1 | grep -i "C= 64" | grep attnv | tee -a $O/fused.txt
done
done
cp tools/xp/libs/new.so $L
