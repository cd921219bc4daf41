"""Diagnostic: tools/bench_train.py against another build of the library: python tools/ab_train.py librrnco_hip_<name>.so [bench_train args]"""
import os, sys, runpy
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "real-routing-nco_amd"))
from rrnco_amd import _lib
_lib.LIB_PATH = _lib.LIB_PATH.replace("librrnco_hip.so", sys.argv[1])
sys.argv = ["bench_train.py"] + sys.argv[2:]
runpy.run_path(os.path.join(ROOT, "tools", "bench_train.py"), run_name="__main__")
