#!/bin/bash
# Round-4 evidence set on one MI355X (run through gpurun from the repo root):  tools/round_end_r04.sh <tag>
#   gpurun_out/<tag>/: PMC traffic + rocprofv3 kernel stats of the bench command FIRST (tools/profile_round.sh), so that the bench line
#   recorded right after quotes the traffic file of the SAME sources; then bench.json (+ op breakdown, stage timeline), the kernel
#   microbenches, the phase stamps, the module-network bench lines (regression check: no PMC passes for them this round), smoke.
T=${1:-r04x}; RD=r04; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$T; mkdir -p $O
cd $R
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1 && echo smoke-ok
bash tools/profile_round.sh gpurun_out/$T/prof 3
mkdir -p profiles/$RD && cp $O/prof/traffic.json profiles/$RD/zz_${T}_pmc_traffic.json
python bench.py --breakdown-file $O/op_breakdown.csv --timeline-file $O/stage_timeline.csv > $O/bench.json 2> $O/bench.err && echo bench-ok
python tools/bench_fused.py 20 > $O/bench_fused.txt 2>&1 && echo fused-ok
python tools/bench_deep.py 30 > $O/bench_deep.txt 2>&1 && echo deep-ok
python tools/probes/stamps/run.py > $O/phase_stamps.txt 2>&1 && echo stamps-ok
python tools/bench_frfn.py --what net --no-cpu > $O/frfnnet_bench.json 2> $O/frfnnet.err && echo frfnnet-ok
python tools/bench_frfn.py --what ast --no-cpu > $O/ast_bench.json 2> $O/ast.err && echo ast-ok
find $O -name "*kernel_trace.csv" -delete; find $O -name "*counter_collection.csv" -delete
python tools/bench_conv3.py > $O/bench_conv3.txt 2>&1 && echo conv3-ok
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $O/trace -- python3 $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-kernel-breakdown > $O/trace.log 2>&1 && echo trace-ok
cd $R
python3 tools/trace_gaps.py $(find $O/trace -name "*kernel_trace.csv" | head -1) 3 > $O/chain_gaps.txt 2>&1
find $O -name "*kernel_trace.csv" -delete
