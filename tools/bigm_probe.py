"""A/B of the row count from which the training path's Linear products take the cast-to-bf16 + glds GEMM route (avlen_set_big_m)."""
import sys, os, runpy
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "4")
from avlen_amd import _lib as L
L.lib.avlen_set_big_m(int(sys.argv[1]))
sys.argv = ["bench.py", "--steps", "3", "--no-roofline", "--no-cpu-baseline", "--no-extras"] + sys.argv[2:]
runpy.run_path(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py"), run_name="__main__")
