"""Diagnostic: which PYTHON lines of the training step launch the small torch kernels (aten glue between the HIP kernels).
Groups the device time of every aten op by the innermost rrnco_amd source line on its stack."""
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
B = int(os.environ.get("PB", "512"))
for i in range(2):
    model.training_step(env.generator(B, generator=gen), optimizer=opt, seed=i)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU], with_stack=True) as prof:
    model.training_step(env.generator(B, generator=gen), optimizer=opt, seed=9)
    torch.cuda.synchronize()
by_line = collections.defaultdict(lambda: [0.0, 0, collections.Counter()])
for e in prof.events():
    if not e.name.startswith("aten::") or e.device_time_total <= 0 or e.cpu_parent is not None and e.cpu_parent.name.startswith("aten::"):
        continue
    where = "?"
    for fr in (e.stack or []):
        if "rrnco_amd" in fr or "/bench.py" in fr or "/train.py" in fr:
            where = fr.split("real-routing-nco_amd/")[-1]
            break
    rec = by_line[where]
    rec[0] += e.device_time_total; rec[1] += 1; rec[2][e.name] += 1
tot = sum(v[0] for v in by_line.values())
print(f"aten device time {tot / 1e3:.2f} ms")
for k, v in sorted(by_line.items(), key=lambda kv: -kv[1][0])[:45]:
    print(f"{v[0] / 1e3:8.3f} ms x{v[1]:<5d} {k[:90]:90s} {dict(v[2].most_common(3))}")
