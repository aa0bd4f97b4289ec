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
# module networks (BASELINE configs[3] / configs[4]): the same two PMC passes over THEIR bench command first, then the bench line that quotes them
for W in net ast; do
  ( cd /tmp && export TMPDIR=/tmp && A="$R/tools/bench_frfn.py --what $W --steps 2 --warmup 1 --no-cpu" && \
    timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/prof_$W/fetch -- python3 $A > $O/prof_$W.fetch.log 2>&1 && \
    timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/prof_$W/write -- python3 $A > $O/prof_$W.write.log 2>&1 && \
    timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$W/stats -- python3 $A > $O/prof_$W.stats.log 2>&1 ) && echo pmc-$W-ok
  python3 tools/pmc_traffic.py $O/prof_$W/fetch $O/prof_$W/write 3 $O/prof_$W/traffic.json fixed > $O/prof_$W.traffic.log 2>&1 && cp $O/prof_$W/traffic.json profiles/$RD/zz_${T}_${W}_hbm_counters.json
done
python tools/bench_frfn.py --what net --breakdown-file $O/frfnnet_breakdown.csv > $O/frfnnet_bench.json 2> $O/frfnnet.err && echo frfnnet-ok
python tools/bench_frfn.py --what ast --breakdown-file $O/ast_breakdown.csv > $O/ast_bench.json 2> $O/ast.err && echo ast-ok
python tools/bench_frfn.py --what ast --loss fflMix --no-cpu > $O/ast_fflmix_bench.json 2> $O/ast_fflmix.err && echo ast-fflmix-ok
python tools/bench_losses.py 4 512 > $O/bench_losses.txt 2>&1 && python tools/bench_losses.py 16 256 >> $O/bench_losses.txt 2>&1 && echo losses-ok
python tools/bench_uiqm.py > $O/uiqm_bench.json 2> $O/uiqm.err && echo uiqm-ok
# opt-in mixed precision (UIR_MIXED, BASELINE configs[4]'s "mixed fp16"): its own lines, after and apart from every f32 number above
bash tools/mixed_bench.sh $T/mixed
find $O -name "*kernel_trace.csv" -delete; find $O -name "*counter_collection.csv" -delete
