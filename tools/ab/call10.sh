#!/bin/bash
T=${1:-r05m}; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$T; mkdir -p $O
cd $R && python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1
python bench.py --no-cpu-baseline --steps 10 --warmup 3 --breakdown-file $O/op_breakdown.csv --timeline-file $O/stage_timeline.csv > $O/bench.json 2> $O/bench.err && echo bench-ok
for i in 1 2; do
  for d in 30 15 45; do
    python tools/ab/engine_flag.py BULK_DEFER=$d 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | head -1 | sed "s/^/BULK_DEFER=$d /"
  done
done | tee $O/ab.txt
python tools/ab/skip.py --steps 10 --warmup 3 --rounds 1 base:none bulk:lane=4 side:lane=1 upsall:name='^ups_' L3:name='^encoders\.3\.' L2:name='^(encoders\.2|decoders\.0)\.' L1:name='^(encoders\.1|decoders\.1)\.' enc0:name='^encoders\.0\.' smallbulk:name='(dWof|dPsum|dT|dWkvf|dWo)$' lnbwd:name='ln[12]_bwd' attn:name='attn_(fwd|bwd)$' red:name='\.red' > $O/skip.txt 2>&1; tail -12 $O/skip.txt
