"""bench.py against another build of the library: python tools/bench_lib.py <path-to-.so> [bench args]"""
import os, sys, runpy
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401  (first: the process must use torch's HIP runtime, as bench.py does)
torch.cuda.is_available()
import dmpp_amd as dm
dm.load_library(sys.argv[1])
sys.argv = ["bench.py"] + sys.argv[2:]
runpy.run_path(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py"), run_name="__main__")
