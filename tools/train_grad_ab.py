"""Diagnostic: the REINFORCE step's gradients at the FULL training shape (512 ATSP n=100 instances) with the round-3 backward
kernels (bf16-pipe attention / logits backward, workspace reductions) against the fp32-MFMA / atomic forms, same sampled tours."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "real-routing-nco_amd"))
import torch, bench
from rrnco_amd.envs import ATSPEnv
from rrnco_amd.models.rl import RRNet
dev = torch.device("cuda")
env = ATSPEnv(generator_params=dict(num_loc=100, device=dev), check_solution=False, device=dev)
batch = env.generator(512, generator=torch.Generator(device=dev).manual_seed(7))

def grads(flags):
    for k in ("RR_ATTN_BWD_F32", "RR_LOGIT_BWD_F32", "RR_TRAIN_WS"):
        os.environ.pop(k, None)
    os.environ.update(flags)
    pol, _ = bench.make_policy(dev)
    pol.train()
    model = RRNet(env, policy=pol)
    out = model.training_step(batch, seed=3)
    g = {n: p.grad.detach().clone() for n, p in pol.named_parameters() if p.grad is not None}
    return out, g

o1, g1 = grads({})
o2, g2 = grads({"RR_ATTN_BWD_F32": "1", "RR_LOGIT_BWD_F32": "1", "RR_TRAIN_WS": "0"})
print("same tours:", bool(torch.equal(o1["actions"], o2["actions"])), " loss", float(o1["loss"]), float(o2["loss"]))
tot = torch.sqrt(sum((g2[n] ** 2).sum() for n in g2))
worst = []
for n in g2:
    d = float((g1[n] - g2[n]).norm() / (g2[n].norm() + 1e-30))
    worst.append((d, n))
worst.sort(reverse=True)
print("relative difference of the whole gradient:", float(torch.sqrt(sum(((g1[n] - g2[n]) ** 2).sum() for n in g2)) / tot))
for d, n in worst[:6]:
    print(f"  {d:.2e}  {n}")
