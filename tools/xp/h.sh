#!/bin/bash
O=gpurun_out/r04h; mkdir -p $O
python -m pytest tests/test_hip_kernels.py tests/test_hip_model.py -m gpu -x -q -k "tall_skinny or gemm or spectral or model or ups or step" > $O/pytest.txt 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest.txt
for m in 0 1; do UIR_FUSED_SPLIT=$m python tools/bench_deep.py 30 dgate > $O/deep_$m.txt 2>&1; done
paste -d'|' <(grep -v amdgpu $O/deep_0.txt | cut -c1-76) <(grep -v amdgpu $O/deep_1.txt | cut -c46-76)
b() { python bench.py --no-cpu-baseline --no-kernel-breakdown --steps 10 --warmup 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['ms_per_step'],3), round(d['value'],1))"; }
b ts; b ts
