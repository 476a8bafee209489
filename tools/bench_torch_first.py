import sys, runpy
import torch  # noqa: F401  (maps torch's bundled ROCm runtime before libydsort is loaded)
sys.argv = ["bench.py"] + sys.argv[1:]
runpy.run_path("bench.py", run_name="__main__")
