#!/bin/bash
O=gpurun_out/r04f; mkdir -p $O
for m in 0 1 2 3; do echo "UIR_XP_NT=$m"; UIR_XP_NT=$m python tools/bench_fused.py 20 dwln 2>&1 | grep dwln; done
b() { python bench.py --no-cpu-baseline --no-kernel-breakdown --steps 10 --warmup 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['ms_per_step'],3), round(d['value'],1))"; }
UIR_XP_NT=0 b nt0; UIR_XP_NT=1 b nt1; UIR_XP_NT=3 b nt3; UIR_XP_NT=0 b nt0; UIR_XP_NT=1 b nt1
