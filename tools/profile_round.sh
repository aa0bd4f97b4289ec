#!/bin/bash
# Round-end measurement set on one MI355X (run through gpurun from the repo root):
#   1. rocprofv3 --kernel-trace --stats of the bench command         -> $OUT/stats
#   2. two separate --pmc passes (FETCH_SIZE, WRITE_SIZE) of the same -> $OUT/traffic.json (tools/pmc_traffic.py)
# usage: tools/profile_round.sh <outdir under gpurun_out> [steps]
OUT=$GRAFT_REPO_ROOT/$1; STEPS=${2:-3}; R=$GRAFT_REPO_ROOT
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="$R/bench.py --steps $STEPS --warmup 1 --no-cpu-baseline --no-kernel-breakdown"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $ARGS > $OUT/stats.log 2>&1 && echo stats-ok
timeout -k 10 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 $ARGS > $OUT/fetch.log 2>&1 && echo fetch-ok
timeout -k 10 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 $ARGS > $OUT/write.log 2>&1 && echo write-ok
python3 $R/tools/pmc_traffic.py $OUT/fetch $OUT/write $((STEPS + 1)) $OUT/traffic.json
