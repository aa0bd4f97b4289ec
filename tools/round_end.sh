#!/bin/bash
# Round-end evidence set on one MI355X (run through gpurun from the repo root):  tools/round_end.sh <tag> [round dir, default r03]
#   gpurun_out/<tag>/  PMC traffic + rocprofv3 kernel stats of the bench command (tools/profile_round.sh) FIRST, so that the bench line
#   recorded right after quotes the traffic file of the SAME sources; then bench.json (+ op breakdown, stage timeline), the kernel
#   microbenches, the module-network bench lines, smoke.  Copy what is to be judged from gpurun_out/<tag>/ into profiles/<round>/.
T=${1:-r03x}; RD=${2:-r03}; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$T; mkdir -p $O
cd $R
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1 && echo smoke-ok
bash tools/profile_round.sh gpurun_out/$T/prof 3
# the traffic file where bench.py looks for it (profiles/r*/...pmc_traffic*.json; newest name sorts last), on THIS box's copy of the tree
mkdir -p profiles/$RD && cp $O/prof/traffic.json profiles/$RD/zz_${T}_pmc_traffic.json
python bench.py --breakdown-file $O/op_breakdown.csv --timeline-file $O/stage_timeline.csv > $O/bench.json 2> $O/bench.err && echo bench-ok
python tools/bench_dwln.py 20 > $O/bench_dwln.txt 2>&1 && echo dwln-ok
python tools/bench_deep.py 30 > $O/bench_deep.txt 2>&1 && echo deep-ok
python tools/bench_frfn.py --what net > $O/frfnnet_bench.json 2> $O/frfnnet.err && echo frfnnet-ok
python tools/bench_frfn.py --what ast > $O/ast_bench.json 2> $O/ast.err && echo ast-ok
python tools/bench_uiqm.py > $O/uiqm_bench.json 2> $O/uiqm.err && echo uiqm-ok
find $O/prof -name "*kernel_trace.csv" -delete; find $O/prof -name "*counter_collection.csv" -delete
