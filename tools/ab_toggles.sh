# usage (inside one gpurun call): bash tools/ab_toggles.sh "VAR=a" "VAR=b" ...   -- one bench run per argument, same device
run() { env "$@" timeout -k 10 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-breakdown 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$*', d['ms_per_step'])"; }
for cfg in "$@"; do run $cfg; done
