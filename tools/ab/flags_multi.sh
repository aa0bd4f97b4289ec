#!/bin/bash
# one GPU call: build, then N rounds of alternating bench.py runs, one per arm.  usage: tools/ab/flags_multi.sh <tag> <rounds> "<arm>" "<arm>" ...
#   an arm is a space-separated list of engine constants (NAME=VALUE, tools/ab/engine_flag.py) and environment variables (ENV:NAME=VALUE); "-" = defaults
T=$1; N=$2; shift 2; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$T; mkdir -p $O
cd $R && python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || { tail -20 $O/build.log; exit 1; }
for i in $(seq 1 $N); do
  for arm in "$@"; do
    fl=""; ev=""
    for w in $arm; do case $w in -) ;; ENV:*) ev="$ev ${w#ENV:}";; *) fl="$fl $w";; esac; done
    ms=$(env $ev timeout -k 10 240 python tools/ab/engine_flag.py $fl 2>>$O/err.txt | grep -o '"ms_per_step": [0-9.]*' | head -1)
    echo "[$arm] $ms" | tee -a $O/ab.txt
  done
done
