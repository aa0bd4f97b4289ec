#!/bin/bash
b() { python bench.py --no-cpu-baseline --no-kernel-breakdown --steps 10 --warmup 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['ms_per_step'],3), round(d['value'],1))"; }
b base; UIR_XP_SKIPBULK=1 b skip-bulk-lane; UIR_XP_SKIPBULK=2 b skip-bulk-grams; b base
