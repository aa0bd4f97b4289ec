#!/bin/bash
O=gpurun_out/r04an; mkdir -p $O
UIR_FORCE_ALLREDUCE=1 python bench.py --no-cpu-baseline --no-kernel-breakdown --steps 5 --warmup 2 > $O/bench_allreduce.json 2> $O/err.txt; echo "rc=$? lines=$(wc -l < $O/bench_allreduce.json)"; cut -c1-120 $O/bench_allreduce.json
python bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_plain.json 2> $O/err3.txt; echo "rc=$? lines=$(wc -l < $O/bench_plain.json)"; python3 -c "import json; d=json.loads(open('$O/bench_plain.json').read()); print(d['value'], d['roofline']['traffic'])"
