"""Times rr_nab_dur (DistAngleFusion with the duration matrix, one encoder layer, row + col block) on C4-shaped inputs:
Bp = 2048 augmented instances, N = 101 nodes.  RR_NABDUR_VARIANT selects the kernel generation."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "real-routing-nco_amd"))
import torch
from rrnco_amd import _lib as L
from rrnco_amd.models import RRNetPolicy

dev = torch.device("cuda")
torch.manual_seed(1234)
pol = RRNetPolicy(env_name="rcvrptw", embed_dim=128, num_heads=8, num_encoder_layers=6, normalization="instance",
                  use_graph_context=False, nab_type="gating", init_embedding_kwargs=dict(sample_size=25)).to(dev).eval()
packed = pol.packed(dev)
Bp, N = int(os.environ.get("BP", 2048)), int(os.environ.get("N", 101))
g = torch.Generator(device=dev).manual_seed(3)
D = torch.rand(Bp, N, N, device=dev, generator=g); T = torch.rand(Bp, N, N, device=dev, generator=g)
locs = torch.rand(Bp, N, 2, device=dev, generator=g)
bias = torch.empty(Bp, 2, N * N, device=dev)
for variant in os.environ.get("VARIANTS", "1,2").split(","):
    os.environ["RR_NABDUR_VARIANT"] = variant
    for rep in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for nr, nc in packed["nabdur"]:
            L.check(L.lib().rr_nab_dur(nr, nc, L.ptr(D), L.ptr(T), L.ptr(locs), L.ptr(bias), Bp, N, L.stream()), "rr_nab_dur")
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / len(packed["nabdur"])
    print(f"variant {variant}: {dt * 1e3:.3f} ms per layer (Bp={Bp}, N={N}), checksum {float(bias.double().sum()):.6f}")
