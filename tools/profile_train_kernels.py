"""Diagnostic: the kernels of one ATSP REINFORCE step that are NOT the library's (torch elementwise / copy / reduction kernels, the fused
optimizer) by GPU time, and the split own / other — device events only, no double counting of operator ranges."""
import os, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "real-routing-nco_amd"))
import torch, bench
from rrnco_amd.envs import ATSPEnv
from rrnco_amd.models.rl import RRNet
from torch.profiler import profile, ProfilerActivity
dev = torch.device("cuda")
pol, w = bench.make_policy(dev)
env = ATSPEnv(generator_params=dict(num_loc=100, device=dev), check_solution=False, device=dev)
pol.train()
model = RRNet(env, policy=pol)
opt = torch.optim.Adam(pol.parameters(), lr=1e-4, fused=True)
gen = torch.Generator(device=dev).manual_seed(1)
for i in range(2):
    model.training_step(env.generator(512, generator=gen), optimizer=opt, seed=i)
batch = env.generator(512, generator=gen)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as prof:
    model.training_step(batch, optimizer=opt, seed=9)
    torch.cuda.synchronize()
agg = collections.defaultdict(lambda: [0.0, 0])
for e in prof.events():
    if e.device_type == torch.autograd.DeviceType.CUDA:
        a = agg[e.name[:120]]; a[0] += e.device_time_total if hasattr(e, "device_time_total") else e.cuda_time_total; a[1] += 1
own = sum(t for n, (t, c) in agg.items() if n.startswith("k_") or n.startswith("void k_"))
oth = {n: v for n, v in agg.items() if not (n.startswith("k_") or n.startswith("void k_"))}
print(f"own kernels {own/1e3:.2f} ms; other {sum(t for t, c in oth.values())/1e3:.2f} ms in {sum(c for t, c in oth.values())} launches")
for n, (t, c) in sorted(oth.items(), key=lambda x: -x[1][0])[:40]:
    print(f"{t/1e3:8.3f} ms x{c:<4d} {n}")
