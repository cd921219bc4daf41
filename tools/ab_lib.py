"""Diagnostic: run bench.py's timed loop against another build of the library (same box A/B):
python tools/ab_lib.py librrnco_hip_prev.so [bench args]."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "real-routing-nco_amd"))
from rrnco_amd import _lib
_lib.LIB_PATH = _lib.LIB_PATH.replace("librrnco_hip.so", sys.argv[1])
sys.argv = ["bench.py"] + sys.argv[2:]
import bench
bench.main()
