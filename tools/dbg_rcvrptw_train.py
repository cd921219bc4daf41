"""Diagnostic: the first steps of train.py --problem rcvrptw, one line per step (loss, gradient norm, range-guard flags, max |w|)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "real-routing-nco_amd"))
import torch
from rrnco_amd.envs import RMTVRPEnv
from rrnco_amd.models import RRNetPolicy
from rrnco_amd.models.rl import RRNet
dev = torch.device("cuda")
torch.manual_seed(1234)
n = 100
policy = RRNetPolicy(env_name="rcvrptw", embed_dim=128, num_heads=8, num_encoder_layers=6, normalization="instance",
                     use_graph_context=False, nab_type="gating", init_embedding_kwargs=dict(sample_size=25)).to(dev)
env = RMTVRPEnv(generator_params=dict(num_loc=n, device=dev), device=dev)
model = RRNet(env, policy=policy, num_augment=8, augment_fn="dihedral8", no_aug_coords=False)
fused = os.environ.get("DBG_FUSED", "1") == "1"
opt = torch.optim.Adam(policy.parameters(), lr=float(os.environ.get("DBG_LR", "4e-4")), weight_decay=1e-6, fused=fused)
gen = torch.Generator(device=dev).manual_seed(1234)
policy.train()
for it in range(int(os.environ.get("DBG_STEPS", "14"))):
    batch = env.generator(64, generator=gen)
    out = model.training_step(batch, optimizer=None, world=1, grad_clip=None, seed=1234 + it)
    raw_bad = [n_ for n_, p in policy.named_parameters() if p.grad is not None and not torch.isfinite(p.grad).all()]
    if raw_bad:
        import collections
        print(f"step {it}: RAW non-finite gradients in {len(raw_bad)} tensors:", collections.Counter(".".join(g.split(".")[:5]) for g in raw_bad).most_common(20))
        print("   ", raw_bad[:30])
        print("    ll finite:", bool(torch.isfinite(out["log_likelihood"]).all()), "replay ll finite:", bool(torch.isfinite(out["replay_log_likelihood"]).all()),
              "reward finite:", bool(torch.isfinite(out["reward"]).all()), "T", out["actions"].shape[1])
        torch.save({"batch": {k: v.cpu() for k, v in batch.items()}, "state_dict": {k: v.cpu() for k, v in policy.state_dict().items()}, "seed": 1234 + it},
                   os.path.join(ROOT, "gpurun_out", "rcvrptw_nan_case.pt"))
        break
    gn = torch.linalg.vector_norm(torch.stack([p.grad.float().norm() for p in policy.parameters() if p.grad is not None]))
    torch._foreach_mul_([p.grad for p in policy.parameters() if p.grad is not None], torch.clamp(1.0 / (gn + 1e-6), max=1.0))
    opt.step(); policy.invalidate_pack()
    out["grad_norm"] = gn
    wmax = max(float(p.detach().abs().max()) for p in policy.parameters())
    bad = [n_ for n_, p in policy.named_parameters() if not torch.isfinite(p).all()]
    gbad = [n_ for n_, p in policy.named_parameters() if p.grad is not None and not torch.isfinite(p.grad).all()]
    if not gbad and it % 10 and it > 3:
        continue
    print(f"step {it}: loss {float(out['loss']):.4f} reward {float(out['reward'].mean()):.4f} grad_norm {float(out['grad_norm']):.3f} "
          f"flags {getattr(policy, 'last_range_flags', None)} max|w| {wmax:.3f} T {out['actions'].shape[1]} non-finite params {len(bad)} {bad[:3]} grads {len(gbad)} {gbad[:12]}", flush=True)
    if gbad:
        import collections
        print("   non-finite gradient tensors by prefix:", collections.Counter(".".join(g.split(".")[:4]) for g in gbad).most_common(12))
        break
