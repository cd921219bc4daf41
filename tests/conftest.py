import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "real-routing-nco_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a ROCm GPU (run on the MI355X box)")


def pytest_sessionstart(session):
    import torch
    torch.set_num_threads(min(8, torch.get_num_threads()))   # the oracle's small CPU ops crawl with hundreds of threads
