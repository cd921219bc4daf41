"""Diagnostic: where the autograd replay of the training step spends its time (encoder vs decoder, forward vs backward)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "real-routing-nco_amd"))
import torch
import bench
from rrnco_amd.envs import ATSPEnv, ATSPGenerator
from rrnco_amd.models import grad_replay as G
from rrnco_amd.models.encoder import ATSPInitEmbedding
dev = torch.device("cuda")
pol, w = bench.make_policy(dev); pol.train()
env = ATSPEnv(generator_params=dict(num_loc=100, device=dev), check_solution=False, device=dev)
B, S, N = int(os.environ.get("PB", "128")), 100, 100
td = env.reset(ATSPGenerator(num_loc=100, device=dev)(B, generator=torch.Generator(device=dev).manual_seed(1)))
sidx = ATSPInitEmbedding.sample_indices(td["distance_matrix"], 25)
P = dict(pol.named_parameters())
acts = torch.stack([torch.randperm(N, device=dev) for _ in range(B * S)]).view(B, S, N)
D, locs = td["distance_matrix"], td["locs"].float()
def t(f, n=2):
    f(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
def enc_fwd():
    with torch.no_grad(): G.encode(P, locs, D, sidx, 6, use_checkpoint=False)
def enc_fb():
    pol.zero_grad(); r, c = G.encode(P, locs, D, sidx, 6); (r.sum() + c.sum()).backward()
row, col = [x.detach().requires_grad_() for x in G.encode(P, locs, D, sidx, 6, use_checkpoint=False)]
def dec_fwd():
    with torch.no_grad():
        for a in range(0, B, 64): G.decode_log_likelihood(P, row[a:a+64], col[a:a+64], D[a:a+64], acts[a:a+64])
def dec_fb():
    pol.zero_grad()
    for a in range(0, B, 64): G.decode_log_likelihood(P, row[a:a+64], col[a:a+64], D[a:a+64], acts[a:a+64]).sum().backward()
print(f"B={B}: encoder fwd {t(enc_fwd):.0f} ms, fwd+bwd(ckpt) {t(enc_fb):.0f} ms; decoder fwd {t(dec_fwd):.0f} ms, fwd+bwd {t(dec_fb):.0f} ms")
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CUDA]) as prof:
    enc_fb(); dec_fb(); torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=14, max_name_column_width=70))
