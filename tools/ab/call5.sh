#!/bin/bash
# state after conv3b: op breakdown + stage timeline + timing-only skip arms
T=${1:-r05e}; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$T; mkdir -p $O
cd $R && python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1
python bench.py --no-cpu-baseline --steps 10 --warmup 3 --breakdown-file $O/op_breakdown.csv --timeline-file $O/stage_timeline.csv > $O/bench.json 2> $O/bench.err && echo bench-ok
python tools/ab/skip.py --steps 10 --warmup 3 --rounds 2 base:none bulk:lane=4 supsall:name='Sups' shuffles:name='Sups.shuffle' upsall:name='^ups_' dft:name='dft' fuse:name='_fuse\.' \
   L3:name='^encoders\.3\.' L2:name='^(encoders\.2|decoders\.0)\.' L2bulk:name='^(encoders\.2|decoders\.0)\..*(dW|red_batch)' grams:name='dW(in|out|qkv)$' dwwgrad:name='dw\.dW$' redbatch:name='red_batch' > $O/skip.txt 2>&1 && echo skip-ok
cat $O/skip.txt | tail -14
