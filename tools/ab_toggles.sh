run() { env "$@" timeout -k 10 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-breakdown 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$*', d['ms_per_step'])"; }
run A=1
run UIR_GDFN_FUSED=16,32,64
run UIR_QKV_FUSED=16,32,64
run UIR_ATTNV_FUSED=16,32,64
run UIR_ATTNV_FUSED=32,64,128
run A=2
