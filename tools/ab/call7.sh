#!/bin/bash
# gramb under the profiler: per-kernel durations of the deep-level Gram microbench, then LDS / VALU counters
T=${1:-r05i}; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$T; mkdir -p $O
cd $R && python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 $R/tools/bench_deep.py 10 Gram > $O/trace.log 2>&1 && echo trace-ok
find $O/trace -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats.csv
find $O/trace -name "*kernel_trace.csv" -delete
timeout -k 10 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $O/pmc1 -- python3 $R/tools/bench_deep.py 3 Gram > $O/pmc1.log 2>&1 && echo pmc1-ok
timeout -k 10 300 rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM --output-format csv -d $O/pmc2 -- python3 $R/tools/bench_deep.py 3 Gram > $O/pmc2.log 2>&1 && echo pmc2-ok
cd $R
python3 - <<'PY' > $O/pmc_summary.txt 2>&1
import csv, glob, collections, os, sys
O=os.environ.get("O") or sys.argv[-1]
PY
for d in pmc1 pmc2; do f=$(find $O/$d -name "*counter_collection.csv" | head -1); python3 -c "
import csv,collections,sys
agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
for r in csv.DictReader(open('$f')):
    k=r['Kernel_Name'][:60]
    agg[k][r['Counter_Name']]+=float(r['Counter_Value']); 
for k,v in agg.items():
    print(k, {c: round(x) for c,x in v.items()})
" >> $O/pmc_summary.txt 2>&1; done
find $O -name "*counter_collection.csv" -delete
cat $O/kernel_stats.csv | cut -d, -f1-8 | head -12
cat $O/pmc_summary.txt | head -30
