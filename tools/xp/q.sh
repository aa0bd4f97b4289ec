#!/bin/bash
O=gpurun_out/r04aj; mkdir -p $O
python -m pytest tests/test_frfn_net.py tests/test_ast.py tests/test_module_fullsize.py tests/test_decoder_block.py tests/test_encoder_block.py tests/test_frfn.py -q -x > $O/pytest.txt 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.txt
run() { PYTHONPATH=. python3 tools/xp/ab_frfn.py "$@" 2>$O/err.txt | grep '^{' | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$*', round(d['ms_per_step'],3), round(d['value'],1))" | tee -a $O/ab.txt; }
for i in 1 2; do
  run net
  run ast
done
