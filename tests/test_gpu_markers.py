"""`-m gpu`: the opt-in rocprofv3 range markers (RR_MARKERS=1, rrnco_amd/_lib.py): every launcher call bracketed by a roctx range named
after the SURVEY section-3 kernel group (K1 .. K10) it replaces.  Without a profiler attached the ranges are no-ops: the wrapped library
must run the smoke rollout unchanged.  (The trace itself: tools/marker_trace.sh -> profiles/r05/markers_*.)"""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_marked_library_runs_the_smoke_rollout():
    code = ("import __graft_entry__ as g; from rrnco_amd import _lib as L; lib = L.lib(); "
            "assert hasattr(lib.rr_rollout, '__wrapped__') and hasattr(lib.rr_init_embed, '__wrapped__'); g.smoke(); print('marked ok')")
    env = dict(os.environ, RR_MARKERS="1", PYTHONPATH=ROOT + os.pathsep + os.path.join(ROOT, "real-routing-nco_amd"))
    out = subprocess.run([sys.executable, "-c", code], cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    assert "marked ok" in out.stdout
