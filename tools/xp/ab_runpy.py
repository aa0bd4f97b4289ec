import sys, runpy
import underwater_image_restoration_amd.engine as e
sys.argv = ["bench.py", "--no-cpu-baseline", "--no-kernel-breakdown", "--steps", "3", "--warmup", "1"]
runpy.run_path("bench.py", run_name="__main__")
