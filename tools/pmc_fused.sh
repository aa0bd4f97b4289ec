#!/bin/bash
# PMC passes over the headline step for its tile-fused kernels (separate passes, kernel-trace only; MI355X_MICROARCH.md)
# usage: [PMC_REGEX=...] tools/pmc_fused.sh <outdir>   -> <outdir>/summary.txt   (PMC_REGEX: kernels to collect, default the five fused families)
R=$GRAFT_REPO_ROOT; OUT=$R/$1
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run() { # name counters...
  n=$1; shift
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc "$@" --kernel-include-regex "${PMC_REGEX:-dwln_bwd|gdfn_fwd|qkv_fwd|attnv_fwd|attnv_bwd}" --output-format csv -d $OUT/$n -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-breakdown > $OUT/$n.log 2>&1
}
run sq1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA
run sq2 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_VALU_MFMA_BUSY_CYCLES
run sq3 SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU GRBM_GUI_ACTIVE
find $OUT -name "*kernel_trace.csv" -delete
python3 $R/tools/pmc_fused_sum.py $OUT > $OUT/summary.txt 2>&1
