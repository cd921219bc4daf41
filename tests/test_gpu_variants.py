"""`-m gpu`: the kernel generations still compiled into the library behind environment switches ("each either
gets one agreement test or leaves the shipped .so").  k_rollout and k_enc_block (first generations) left the library in round 4.
What remains switchable: the first-generation env.step kernels (RR_STEP_VARIANT=0: also the path odd shapes take) and the
one-wave-per-row selection kernel (RR_SELECT_VARIANT=0: also the path of the top-k / top-p filters); the three duration-NAB
generations have their own test (test_gpu_rcvrptw.py::test_duration_nab_three_kernel_generations_agree).  The switches are read once
per process, so the step-wise parity tests of all three problems are re-run in a child process with the old generations selected:
the same golden tours of the real reference must come out."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("switch", ["RR_STEP_VARIANT", "RR_SELECT_VARIANT"])
def test_first_generation_step_and_select_kernels_reproduce_the_golden_tours(switch):
    env = dict(os.environ, **{switch: "0"})
    # fused=False: the reference's own decode loop (decoder.forward, process_logits + select, env.step as separate launches per step)
    cmd = [sys.executable, "-m", "pytest", "-q", "-x", "-m", "gpu", "-p", "no:cacheprovider",
           os.path.join(ROOT, "tests", "test_gpu_atsp.py"), os.path.join(ROOT, "tests", "test_gpu_rcvrp.py"), os.path.join(ROOT, "tests", "test_gpu_rcvrptw.py"),
           "-k", "(match_reference and False) or env_step or stepwise or step_and or select_kernel or evaluate_mode"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=1500, env=env, cwd=ROOT)
    tail = (r.stdout + r.stderr)[-2500:]
    assert r.returncode == 0, tail
    assert " passed" in r.stdout and "failed" not in r.stdout, tail
    n = int(r.stdout.strip().splitlines()[-1].split(" passed")[0].split()[-1])
    assert n >= 10, tail                                    # the selection really matched the step-wise parity tests
