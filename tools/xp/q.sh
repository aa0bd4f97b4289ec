#!/bin/bash
O=gpurun_out/r04ah; mkdir -p $O
python -m pytest tests/test_frfn_net.py tests/test_ast.py tests/test_module_fullsize.py tests/test_decoder_block.py tests/test_encoder_block.py tests/test_frfn.py -q > $O/pytest.txt 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest.txt
