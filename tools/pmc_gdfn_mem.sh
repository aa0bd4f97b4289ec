#!/bin/bash
# memory-side PMC passes for the fused GDFN microbench
OUT=$1; R=$GRAFT_REPO_ROOT
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run() { n=$1; shift
  ONLY=32x256 REPS=3 timeout -k 10 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$n -- python3 $R/tools/bench_gdfn.py > $OUT/$n.log 2>&1
}
run tcc1 FETCH_SIZE GRBM_GUI_ACTIVE
run tcc2 WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
# the four TA counters in ONE pass exceed what rocprofv3 can program (round 1: 'error code 38: Request exceeds the capabilities of the
# hardware'): one derived TA sum per pass
run ta1 TA_TA_BUSY_sum
run ta2 TA_ADDR_STALLED_BY_TC_CYCLES_sum
run ta3 TA_DATA_STALLED_BY_TC_CYCLES_sum
run tcc3 TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_STALL_sum TCC_WRITE_sum
run tcp TCP_PENDING_STALL_CYCLES_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_READ_REQ_sum TCP_TCR_TCP_STALL_CYCLES_sum
