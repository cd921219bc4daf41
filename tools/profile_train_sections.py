"""Diagnostic: wall time of the sections of one ATSP REINFORCE step (BASELINE configs[4] shape), each bracketed by device synchronisations
(so host-bound sections show their host time; nested sections are listed under their own name as well)."""
import os, sys, time, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "real-routing-nco_amd"))
import torch, bench
from rrnco_amd.envs import ATSPEnv
from rrnco_amd.models.rl import RRNet
import rrnco_amd.models.rl as RL, rrnco_amd.models.grad_replay as GR, rrnco_amd.models.dec_backward as DB, rrnco_amd.models.enc_backward as EB
import rrnco_amd.models.policy as PO
dev = torch.device("cuda")
pol, w = bench.make_policy(dev)
env = ATSPEnv(generator_params=dict(num_loc=100, device=dev), check_solution=False, device=dev)
pol.train()
model = RRNet(env, policy=pol)
opt = torch.optim.Adam(pol.parameters(), lr=1e-4, fused=True)
gen = torch.Generator(device=dev).manual_seed(1)
T = collections.OrderedDict(); depth = [0]
def wrap(mod, name, label=None):
    f = getattr(mod, name); label = label or name
    def g(*a, **k):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        r = f(*a, **k)
        torch.cuda.synchronize(); T[label] = T.get(label, 0.0) + time.perf_counter() - t0
        return r
    setattr(mod, name, g)
wrap(DB, "decoder_backward"); wrap(EB, "encoder_backward"); wrap(EB, "train_packs"); wrap(EB, "nab_grad_from_hist")
wrap(torch.autograd, "backward", "torch.autograd.backward")
wrap(RL, "allreduce_flat_gradients"); wrap(RL, "reinforce_loss")
wrap(PO.RRNetPolicy, "forward", "policy.forward"); wrap(PO.RRNetPolicy, "packed", "policy.packed"); wrap(PO.RRNetPolicy, "invalidate_pack")
wrap(opt, "step", "optimizer.step")
wrap(env, "reset", "env.reset")
batches = [env.generator(512, generator=gen) for _ in range(5)]
for i in range(2):
    model.training_step(batches[i], optimizer=opt, seed=i)
T.clear()
torch.cuda.synchronize(); t0 = time.perf_counter()
for i in range(3):
    model.training_step(batches[2 + i], optimizer=opt, seed=9 + i)
torch.cuda.synchronize(); tot = (time.perf_counter() - t0) / 3
print(f"step (serialised by the section syncs): {tot*1e3:.2f} ms")
for k, v in T.items():
    print(f"{v/3*1e3:8.2f} ms  {k}")
