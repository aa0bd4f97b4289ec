#!/bin/bash
T=${1:-r05o}; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$T; mkdir -p $O
cd $R && python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1
for i in 1 2 3; do
  for v in 0 16384 65536 262144; do
    python tools/ab/engine_flag.py FUPS_PHA_SIDE_MAXROWS=$v 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | head -1 | sed "s/^/pha side <= $v /"
  done
done | tee $O/ab.txt
timeout -k 10 900 python -m pytest tests/test_hip_model.py -x -q -m gpu > $O/pytest_model.txt 2>&1; echo "pytest model rc=$?"; tail -3 $O/pytest_model.txt
python tools/ab/skip.py --steps 10 --warmup 3 --rounds 1 base:none grams:name='\.dW(in|qkv)$|^(enc|dec).*[0-9]\.dWout$' dwwgrad:name='dw\.dW$' chainred:name='gram\.red$|dWAB\.red$' fupsmlp:name='_fuse\.[02](\.dX)?$' smallconv:name='^(outputl|output|down[123]\.conv|embed)(\.dX)?$|im2col' > $O/skip.txt 2>&1; tail -6 $O/skip.txt
