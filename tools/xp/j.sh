#!/bin/bash
O=gpurun_out/r04j; mkdir -p $O
python -m pytest tests/test_module_fullsize.py tests/test_hip_model.py -m gpu -x -q -k "free_memory or private_streams" > $O/pytest_new.txt 2>&1; echo "new tests rc=$?"; tail -4 $O/pytest_new.txt
python -m pytest tests/test_ast.py -m gpu -x -q -k "cfg5" -s > $O/pytest_cfg5.txt 2>&1; echo "cfg5 rc=$?"; grep "cfg5 fp16" $O/pytest_cfg5.txt; tail -3 $O/pytest_cfg5.txt
