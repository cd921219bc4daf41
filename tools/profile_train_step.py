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
with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as prof:
    model.training_step(ATSPGenerator(num_loc=100, device=dev)(B, generator=gen), optimizer=opt, seed=9)
    torch.cuda.synchronize()
cpu = sorted([e for e in prof.key_averages() if e.key.startswith("aten::") or e.key.startswith("hip")], key=lambda e: -e.count)
if os.environ.get("PROFILE_HOST"):
    print("host-side ops by call count:")
    for e in cpu[:25]:
        print(f"  x{e.count:<6d} {e.self_cpu_time_total / 1e3:8.2f} ms self-cpu  {e.key[:80]}")
ev = sorted([e for e in prof.key_averages() if e.self_device_time_total > 0], key=lambda e: -e.self_device_time_total)
tot = sum(e.self_device_time_total for e in ev)
print(f"GPU time of one training step: {tot / 1e3:.2f} ms over {sum(e.count for e in ev)} launches")
acc = 0.0
for e in ev[:40]:
    acc += e.self_device_time_total
    print(f"{e.self_device_time_total / 1e3:9.3f} ms {100 * e.self_device_time_total / tot:5.1f}% cum {100 * acc / tot:5.1f}%  x{e.count:<4d} {e.key[:110]}")
