"""Diagnostic: random ATSP / RCVRP / RCVRPTW shapes through tests/test_gpu_shapes.py's live-oracle comparison (python tools/fuzz_shapes.py <seed>)."""
import os, sys, random
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "real-routing-nco_amd"))
import torch
sys.path.insert(0, os.path.join(ROOT, "tests"))
from tests import test_gpu_shapes as T
random.seed(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
bad = 0
for it in range(14):
    N = random.choice([5, 9, 16, 17, 31, 32, 33, 47, 48, 64, 65, 80, 96, 100, 103])
    B = random.randint(1, 9)
    S = random.choice([1, 2, N // 2 or 1, N, min(N + 7, 2 * N), 16, 17, 32, 48])
    S = max(1, min(S, N))          # POMO starts are distinct nodes
    ss = max(1, min(25, N - 2))
    try:
        T._run(N, B, S, ss, seed=1000 + it, layers=1)
        print("ok  atsp", N, B, S, ss, flush=True)
    except Exception as e:
        bad += 1
        print("FAIL atsp", N, B, S, ss, repr(e)[:300], flush=True)
for it in range(6):
    prob = random.choice(["rcvrp", "rcvrptw"])
    N = random.choice([8, 15, 24, 40, 63, 77, 100, 102])
    B = random.randint(1, 5)
    S = random.choice([2, N // 2, N, N + 1 if prob == "rcvrp" else N])
    S = max(2, min(S, N + (1 if prob == "rcvrp" else 0)))
    ss = max(1, min(25, N - 2))
    try:
        T._run_vrp(prob, N, B, S, ss, seed=2000 + it, layers=1)
        print("ok ", prob, N, B, S, ss, flush=True)
    except Exception as e:
        bad += 1
        print("FAIL", prob, N, B, S, ss, repr(e)[:300], flush=True)
print("failures:", bad)
