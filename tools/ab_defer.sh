run() { env "$@" timeout -k 10 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-breakdown 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$*', d['ms_per_step'])"; }
run UIR_DEFER=0
run UIR_DEFER=1 "UIR_DEFER_AT=bwd.dec2(L0,C32)"
run UIR_DEFER=1 UIR_DEFER_AT=bwd.ups_3
run UIR_DEFER=1 "UIR_DEFER_AT=bwd.dec1(L1)"
run UIR_DEFER=0
