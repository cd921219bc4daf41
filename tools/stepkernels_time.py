"""Times the per-step kernels of the reference-shaped decode loop at the headline size (R = 4096 x 100 rollouts, N = 100):
rr_select (greedy / sampling) and rr_atsp_step, HIP events around 20 back-to-back launches on rotating buffers."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "real-routing-nco_amd"))
import torch
from rrnco_amd import _lib as L

dev = torch.device("cuda")
R, N, NB = int(os.environ.get("R", 409600)), int(os.environ.get("N", 100)), 3
g = torch.Generator(device=dev).manual_seed(1)
logits = [torch.randn(R, N, device=dev, generator=g) * 3 for _ in range(NB)]
mask = [(torch.rand(R, N, device=dev, generator=g) < 0.6).to(torch.uint8) for _ in range(NB)]
for m in mask: m[:, 0] = 1
sel = torch.empty(R, dtype=torch.int64, device=dev); lp = torch.empty(R, device=dev)
mout = torch.empty(R, N, dtype=torch.uint8, device=dev); done = torch.empty(R, dtype=torch.uint8, device=dev)
act = torch.randint(0, N, (R,), device=dev)


def timed(fn, reps=20):
    fn(0); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for i in range(reps): fn(i % NB)
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


for name, mode in (("greedy", 0), ("sampling", 1)):
    us = timed(lambda i: L.check(L.lib().rr_select(L.ptr(logits[i]), L.ptr(mask[i]), None, L.ptr(sel), L.ptr(lp), None, R, N, 10.0, 1.0,
                                                  mode, 7, 3, 0, 0.0, L.stream()), "rr_select"))
    nbytes = R * (5 * N + 12)
    print(f"rr_select {name}: {us:.1f} us, {nbytes / 1e6:.0f} MB -> {nbytes / us / 1e6:.2f} TB/s ({nbytes / us / 1e6 / 8 * 100:.0f} % of 8 TB/s)")
us = timed(lambda i: L.check(L.lib().rr_atsp_step(L.ptr(act), L.ptr(mask[i]), L.ptr(mout), L.ptr(done), R, N, L.stream()), "rr_atsp_step"))
nbytes = R * (2 * N + 9)
print(f"rr_atsp_step: {us:.1f} us, {nbytes / 1e6:.0f} MB -> {nbytes / us / 1e6:.2f} TB/s ({nbytes / us / 1e6 / 8 * 100:.0f} % of 8 TB/s)")
us = timed(lambda i: mout.copy_(mask[i]))
print(f"reference point, plain device copy of the mask (torch copy_): {us:.1f} us, {2 * R * N / 1e6:.0f} MB -> {2 * R * N / us / 1e6:.2f} TB/s")
lo = torch.empty_like(logits[0])
us = timed(lambda i: lo.copy_(logits[i]))
print(f"reference point, plain device copy of the logits: {us:.1f} us, {8 * R * N / 1e6:.0f} MB -> {8 * R * N / us / 1e6:.2f} TB/s")
