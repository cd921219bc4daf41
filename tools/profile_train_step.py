"""Diagnostic: top GPU kernels of one REINFORCE training step (BASELINE configs[4] shape) by torch.profiler."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "real-routing-nco_amd"))
import torch, bench
from rrnco_amd.envs import ATSPEnv, ATSPGenerator
from rrnco_amd.models.rl import RRNet
from torch.profiler import profile, ProfilerActivity
dev = torch.device("cuda")
PROBLEM = os.environ.get("PROBLEM", "atsp")               # atsp | rcvrp | rcvrptw
if PROBLEM == "atsp":
    pol, w = bench.make_policy(dev)
    env = ATSPEnv(generator_params=dict(num_loc=100, device=dev), check_solution=False, device=dev)
else:
    from rrnco_amd.envs import RCVRPEnv, RMTVRPEnv
    from rrnco_amd.models import RRNetPolicy
    torch.manual_seed(1234)
    pol = RRNetPolicy(env_name=PROBLEM, embed_dim=128, num_heads=8, num_encoder_layers=6, normalization="instance",
                      use_graph_context=False, nab_type="gating", init_embedding_kwargs=dict(sample_size=25)).to(dev)
    gp = dict(num_loc=100, device=dev)
    env = RCVRPEnv(generator_params=gp, check_solution=False, device=dev) if PROBLEM == "rcvrp" else RMTVRPEnv(generator_params=gp, device=dev)
pol.train()
model = RRNet(env, policy=pol)
opt = torch.optim.Adam(pol.parameters(), lr=1e-4, fused=True)
gen = torch.Generator(device=dev).manual_seed(1)
B = int(os.environ.get("PB", "512"))
for i in range(2):
    model.training_step(env.generator(B, generator=gen), optimizer=opt, seed=i)
torch.cuda.synchronize()
batch = env.generator(B, generator=gen)          # outside the profile: the generator's 100 closure passes are not the step's
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as prof:
    model.training_step(batch, optimizer=opt, seed=9)
    torch.cuda.synchronize()
cpu = sorted([e for e in prof.key_averages() if e.key.startswith("aten::") or e.key.startswith("hip")], key=lambda e: -e.count)
if os.environ.get("PROFILE_HOST"):
    print("host-side ops by call count:")
    for e in cpu[:25]:
        print(f"  x{e.count:<6d} {e.self_cpu_time_total / 1e3:8.2f} ms self-cpu  {e.key[:80]}")
ev = sorted([e for e in prof.key_averages() if e.self_device_time_total > 0], key=lambda e: -e.self_device_time_total)
tot = sum(e.self_device_time_total for e in ev)
nk = sum(1 for e in prof.events() if e.device_type == torch.autograd.DeviceType.CUDA)
print(f"GPU time of one training step: {tot / 1e3:.2f} ms; {nk} device kernels / copies launched ({sum(e.count for e in ev)} rows below count "
      "operator ranges too)")
acc = 0.0
for e in ev[:40]:
    acc += e.self_device_time_total
    print(f"{e.self_device_time_total / 1e3:9.3f} ms {100 * e.self_device_time_total / tot:5.1f}% cum {100 * acc / tot:5.1f}%  x{e.count:<4d} {e.key[:110]}")
