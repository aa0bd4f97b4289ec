#!/bin/bash
# PMC passes for the uir_dwln_bwd microbench (separate passes, kernel-trace only; see MI355X_MICROARCH.md)
# usage: tools/pmc_dwln.sh <outdir>
R=$GRAFT_REPO_ROOT; OUT=$R/$1
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run() { # name counters...
  n=$1; shift
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$n -- python3 $R/tools/bench_dwln.py 3 first > $OUT/$n.log 2>&1
}
run sq1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA
run sq2 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_VALU_MFMA_BUSY_CYCLES
run sq3 SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_BRANCH SQ_IFETCH
python3 $R/tools/pmc_summarize.py $OUT dwln > $OUT/summary.txt 2>&1
