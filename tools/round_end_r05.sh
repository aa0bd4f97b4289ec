#!/bin/bash
# Round-5 evidence set on one MI355X (run through gpurun from the repo root):  tools/round_end_r05.sh <tag>
#   gpurun_out/<tag>/: rebuild (stale objects only), smoke, PMC traffic + rocprofv3 kernel stats of the bench command FIRST (tools/profile_round.sh),
#   so that the bench line recorded right after quotes the traffic file of the SAME sources; then bench.json (+ op breakdown, stage timeline), the
#   untraced chain-gap analysis (HIP events), the kernel microbenches of this round's new kernels, the module-network bench lines (regression check).
T=${1:-r05z}; RD=r05; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$T; mkdir -p $O
cd $R && python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1 && echo smoke-ok
bash tools/profile_round.sh gpurun_out/$T/prof 3
mkdir -p profiles/$RD && rm -f profiles/$RD/zz_*_pmc_traffic.json && cp $O/prof/traffic.json profiles/$RD/zz_${T}_pmc_traffic.json   # ONE traffic file per round: bench.py quotes the last in name order
find $O/prof/stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats.csv
python bench.py --breakdown-file $O/op_breakdown.csv --timeline-file $O/stage_timeline.csv > $O/bench.json 2> $O/bench.err && echo bench-ok
python tools/chain_gaps_events.py 12 3 > $O/chain_gaps_events.txt 2>&1 && echo gaps-ok
python tools/bench_conv3b.py --tiles > $O/bench_conv3b.txt 2>&1 && echo conv3b-ok
python tools/bench_deep.py 30 > $O/bench_deep.txt 2>&1 && echo deep-ok
python tools/bench_fused.py 20 > $O/bench_fused.txt 2>&1 && echo fused-ok
python tools/bench_value_bwd.py > $O/bench_value_bwd.txt 2>&1 && echo value-bwd-ok
python tools/bench_frfn.py --what net --no-cpu > $O/frfnnet_bench.json 2> $O/frfnnet.err && echo frfnnet-ok
python tools/bench_frfn.py --what ast --no-cpu > $O/ast_bench.json 2> $O/ast.err && echo ast-ok
find $O -name "*kernel_trace.csv" -delete; find $O -name "*counter_collection.csv" -delete
tail -1 $O/bench.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('ms_per_step', d['ms_per_step'], 'frac', d['roofline']['frac'], 'traffic', d['roofline']['traffic'], 'frac_counter', d['roofline'].get('frac_counter_bytes'))"
