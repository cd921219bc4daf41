"""Diagnostic: top GPU kernels of one REINFORCE training step (BASELINE configs[4] shape) by torch.profiler."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "real-routing-nco_amd"))
import torch, bench
from rrnco_amd.envs import ATSPEnv, ATSPGenerator
from rrnco_amd.models.rl import RRNet
from torch.profiler import profile, ProfilerActivity
dev = torch.device("cuda")
pol, w = bench.make_policy(dev); pol.train()
env = ATSPEnv(generator_params=dict(num_loc=100, device=dev), check_solution=False, device=dev)
model = RRNet(env, policy=pol)
opt = torch.optim.Adam(pol.parameters(), lr=1e-4, fused=True)
gen = torch.Generator(device=dev).manual_seed(1)
B = int(os.environ.get("PB", "512"))
for i in range(2):
    model.training_step(ATSPGenerator(num_loc=100, device=dev)(B, generator=gen), optimizer=opt, seed=i)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CUDA]) as prof:
    model.training_step(ATSPGenerator(num_loc=100, device=dev)(B, generator=gen), optimizer=opt, seed=9)
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=22, max_name_column_width=64))
