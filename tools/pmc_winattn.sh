#!/bin/bash
# PMC passes over the NewBigFRFNModel step, window-attention backward only (separate passes, kernel-trace only; MI355X_MICROARCH.md)
# usage: tools/pmc_winattn.sh <outdir>    then: python tools/pmc_winattn_sum.py <outdir>
R=$GRAFT_REPO_ROOT; OUT=$R/$1
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run() { # name counters...
  n=$1; shift
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc "$@" --kernel-include-regex "winattn_bwd" --output-format csv -d $OUT/$n -- python3 $R/tools/bench_frfn.py --what net --no-cpu --steps 2 --warmup 1 > $OUT/$n.log 2>&1
}
run sq1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA
run sq2 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_VALU_MFMA_BUSY_CYCLES
run sq3 SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU
run tcc1 FETCH_SIZE GRBM_GUI_ACTIVE
run tcc2 WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
run tcp1 TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum
find $OUT -name "*kernel_trace.csv" -delete
