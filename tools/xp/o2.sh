#!/bin/bash
O=gpurun_out/r04o; mkdir -p $O
cat > $O/ab2.py <<'P'
import sys
import torch
torch.cuda.set_device(0)
import bench
import underwater_image_restoration_amd.engine as e
e.FUSE_VALUE_GRAMS = sys.argv[1] == "1"
sys.argv = ["bench.py", "--no-cpu-baseline", "--no-kernel-breakdown", "--steps", "10", "--warmup", "3"]
bench.main()
P
PYTHONPATH=. AMD_LOG_LEVEL=1 python3 $O/ab.py 1 > $O/out_log.txt 2> $O/err_log.txt; grep -v "^\[uir\]" $O/err_log.txt | head -30
echo ---- ab2
PYTHONPATH=. python3 $O/ab2.py 1 > $O/out2.txt 2> $O/err2.txt; tail -5 $O/err2.txt; cat $O/out2.txt | cut -c1-200
