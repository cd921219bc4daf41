"""Run bench.py's hot path against an alternative build of the library (diagnostic A/B only)."""
import os, sys, runpy
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "real-routing-nco_amd"))
from rrnco_amd import _lib
suffix = sys.argv[1]
_lib.LIB_PATH = _lib.LIB_PATH.replace("librrnco_hip.so", f"librrnco_hip{suffix}.so")
sys.argv = ["bench.py"] + sys.argv[2:]
runpy.run_path(os.path.join(ROOT, "bench.py"), run_name="__main__")
