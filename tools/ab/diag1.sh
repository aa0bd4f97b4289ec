#!/bin/bash
# diagnostic call on one MI355X: baseline + timing-only skip arms + a kernel trace with the chain gaps' producers named
T=${1:-r05a}; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$T; mkdir -p $O
cd $R
python tools/ab/skip.py --steps 10 --warmup 3 --rounds 2 base:none bulk:lane=4 supsdw:name='Sups\.dW' supsconv:name='Sups(\.dX)?$' supsall:name='Sups' upsall:name='^ups_' \
   L3:name='^encoders\.3\.' L2:name='^(encoders\.2|decoders\.0)\.' > $O/skip.txt 2>&1 && echo skip-ok
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $O/trace -- python3 $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-kernel-breakdown > $O/trace.log 2>&1 && echo trace-ok
cd $R
python3 tools/trace_gaps.py $(find $O/trace -name "*kernel_trace.csv" | head -1) 3 12 > $O/chain_gaps.txt 2>&1
tail -3 $O/skip.txt
