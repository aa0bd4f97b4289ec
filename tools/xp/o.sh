#!/bin/bash
O=gpurun_out/r04o; mkdir -p $O
PYTHONPATH=. python3 tools/xp/ab_runpy.py > $O/out_log.txt 2> $O/err_log.txt; tail -2 $O/err_log.txt; cut -c1-150 $O/out_log.txt
