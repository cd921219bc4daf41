"""Diagnostic: which lines of rrnco_amd issue the torch ops of one training step (TorchDispatchMode + the Python stack).
python tools/count_train_ops.py [op substring ...]   (default: every op; prints the 40 busiest (op, line) pairs)"""
import collections, os, sys, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "real-routing-nco_amd"))
import torch, bench
from torch.utils._python_dispatch import TorchDispatchMode
from rrnco_amd.envs import ATSPEnv
from rrnco_amd.models.rl import RRNet
dev = torch.device("cuda")
pol, w = bench.make_policy(dev)
env = ATSPEnv(generator_params=dict(num_loc=100, device=dev), check_solution=False, device=dev)
pol.train()
model = RRNet(env, policy=pol)
opt = torch.optim.Adam(pol.parameters(), lr=1e-4, fused=True)
gen = torch.Generator(device=dev).manual_seed(1)
batches = [env.generator(512, generator=gen) for _ in range(3)]
for i in range(2):
    model.training_step(batches[i], optimizer=opt, seed=i, grad_clip=1.0)
want = sys.argv[1:]
counts = collections.Counter()
class Mode(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        if not want or any(s in name for s in want):
            where = "?"
            for fr in reversed(traceback.extract_stack(limit=40)):
                if "rrnco_amd" in fr.filename or fr.filename.endswith("bench.py"):
                    where = f"{os.path.basename(fr.filename)}:{fr.lineno}"
                    break
            counts[(name, where)] += 1
        return func(*args, **(kwargs or {}))
with Mode():
    model.training_step(batches[2], optimizer=opt, seed=9, grad_clip=1.0)
torch.cuda.synchronize()
print("ops dispatched in one step:", sum(counts.values()))
for (name, where), n in counts.most_common(45):
    print(f"  x{n:<5d} {name:45s} {where}")
