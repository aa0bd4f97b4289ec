#!/bin/bash
O=gpurun_out/r04zy; mkdir -p $O
python bench.py --breakdown-file $O/op_breakdown2.csv --timeline-file $O/stage_timeline2.csv > $O/bench2.json 2> $O/bench2.err && echo bench-ok
tail -c 600 $O/bench2.json
