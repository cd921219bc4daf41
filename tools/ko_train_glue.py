"""Diagnostic: what the torch glue of the training step's encoder backward costs, by knocking pieces out (gradients then WRONG):
KO=nab   the NAB fold chain (folded table built with autograd + its backward) replaced by a constant table
KO=init  the init embedding's autograd replay skipped"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "real-routing-nco_amd"))
import torch, bench
from rrnco_amd.envs import ATSPEnv
from rrnco_amd.models.rl import RRNet
from rrnco_amd.models import enc_backward as EB, grad_replay as GR
dev = torch.device("cuda")
pol, w = bench.make_policy(dev)
env = ATSPEnv(generator_params=dict(num_loc=100, device=dev), check_solution=False, device=dev)
pol.train()
model = RRNet(env, policy=pol)
opt = torch.optim.Adam(pol.parameters(), lr=1e-4, fused=True)
gen = torch.Generator(device=dev).manual_seed(1)
def run(tag, n=6):
    for i in range(2):
        model.training_step(env.generator(512, generator=gen), optimizer=opt, seed=i)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(n):
        model.training_step(env.generator(512, generator=gen), optimizer=opt, seed=10 + i)
    torch.cuda.synchronize()
    print(f"{tag:28s} {(time.perf_counter() - t0) / n * 1e3:7.2f} ms per step")
run("all")
orig_tab, orig_hist = EB._nab_tab, EB.nab_grad_from_hist
cache = {}
def const_tab(P, p, alpha):
    if p not in cache:
        with torch.no_grad():
            cache[p] = orig_tab(P, p, alpha).detach()
    return cache[p]
EB._nab_tab = const_tab
EB.nab_grad_from_hist = lambda tabs, hist: torch.zeros_like(tabs)
_bw = torch.autograd.backward
def bw(ts, gs):
    keep = [(t, g) for t, g in zip(ts, gs) if t.requires_grad]
    if keep:
        _bw([t for t, _ in keep], [g for _, g in keep])
torch.autograd.backward = bw
run("without the NAB fold chain")
EB._nab_tab, EB.nab_grad_from_hist = orig_tab, orig_hist
orig_init = GR._init_embedding
def no_init(P, locs, D, sidx):
    z = torch.zeros(D.shape[0], D.shape[-1], 128, device=D.device)
    return z, z
GR._init_embedding = no_init
run("without the init-embedding replay")
