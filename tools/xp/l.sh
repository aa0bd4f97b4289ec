#!/bin/bash
O=gpurun_out/r04l; mkdir -p $O
P=underwater_image_restoration_amd
python -m pytest tests/test_hip_kernels.py -m gpu -x -q -k "dwln or gdfn_fwd or qkv_fwd or attnv" > $O/pytest.txt 2>&1; echo "pytest rc=$?"; tail -2 $O/pytest.txt
b() { python bench.py --no-cpu-baseline --no-kernel-breakdown --steps 10 --warmup 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['ms_per_step'],3), round(d['value'],1))"; }
cp $P/libuir_hip.so /tmp/new.so
python tools/bench_fused.py 20 > $O/fused_new.txt 2>&1; b new
cp $P/libuir_hip_old.so $P/libuir_hip.so
python tools/bench_fused.py 20 > $O/fused_old.txt 2>&1; b old
cp /tmp/new.so $P/libuir_hip.so; b new
cp $P/libuir_hip_old.so $P/libuir_hip.so; b old
cp /tmp/new.so $P/libuir_hip.so
paste -d'|' <(grep -v "amdgpu\|UIR_F" $O/fused_old.txt | cut -c1-64) <(grep -v "amdgpu\|UIR_F" $O/fused_new.txt | cut -c35-64)
