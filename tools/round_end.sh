#!/bin/bash
# Round-end evidence set on one MI355X (run through gpurun from the repo root):  tools/round_end.sh <tag>
#   gpurun_out/<tag>/  bench.json (+ op breakdown, stage timeline), frfn / encoder / decoder / cfg4 net / AST / uiqm bench lines, rocprofv3 kernel stats of the
#   bench command, PMC traffic (tools/profile_round.sh), smoke
T=${1:-r02x}; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$T; mkdir -p $O
cd $R
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1 && echo smoke-ok
python bench.py --breakdown-file $O/op_breakdown.csv --timeline-file $O/stage_timeline.csv > $O/bench.json 2> $O/bench.err && echo bench-ok
python tools/bench_frfn.py > $O/frfn_bench.json 2> $O/frfn.err && echo frfn-ok
python tools/bench_frfn.py --what encoder > $O/encoder_bench.json 2> $O/encoder.err && echo encoder-ok
python tools/bench_frfn.py --what decoder > $O/decoder_bench.json 2> $O/decoder.err && echo decoder-ok
python tools/bench_frfn.py --what net > $O/frfnnet_bench.json 2> $O/frfnnet.err && echo frfnnet-ok
python tools/bench_frfn.py --what ast > $O/ast_bench.json 2> $O/ast.err && echo ast-ok
python tools/bench_uiqm.py > $O/uiqm_bench.json 2> $O/uiqm.err && echo uiqm-ok
bash tools/profile_round.sh gpurun_out/$T/prof 3
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof/uiqm -- python3 $R/tools/bench_uiqm.py --no-cpu > $O/prof/uiqm.log 2>&1 && echo uiqm-prof-ok
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof/encoder -- python3 $R/tools/bench_frfn.py --what encoder --no-cpu > $O/prof/encoder.log 2>&1 && echo encoder-prof-ok
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof/decoder -- python3 $R/tools/bench_frfn.py --what decoder --no-cpu > $O/prof/decoder.log 2>&1 && echo decoder-prof-ok
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof/frfnnet -- python3 $R/tools/bench_frfn.py --what net --no-cpu --steps 5 > $O/prof/frfnnet.log 2>&1 && echo frfnnet-prof-ok
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof/ast -- python3 $R/tools/bench_frfn.py --what ast --no-cpu --steps 5 > $O/prof/ast.log 2>&1 && echo ast-prof-ok
find $O/prof -name "*kernel_trace.csv" -delete; find $O/prof -name "*counter_collection.csv" -delete
