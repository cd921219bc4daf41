"""Diagnostic: replays gpurun_out/rcvrptw_nan_case.pt (tools/dbg_rcvrptw_train.py) with RR_NAN_TRACE=1."""
import os, sys
os.environ["RR_NAN_TRACE"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "real-routing-nco_amd"))
import torch
from rrnco_amd import TensorDict
from rrnco_amd.envs import RMTVRPEnv
from rrnco_amd.models import RRNetPolicy
from rrnco_amd.models.rl import RRNet
case = torch.load(os.path.join(ROOT, "scratch", "rcvrptw_nan_case.pt"), weights_only=False)
dev = torch.device("cuda")
policy = RRNetPolicy(env_name="rcvrptw", embed_dim=128, num_heads=8, num_encoder_layers=6, normalization="instance",
                     use_graph_context=False, nab_type="gating", init_embedding_kwargs=dict(sample_size=25)).to(dev)
policy.load_state_dict(case["state_dict"]); policy.train()
env = RMTVRPEnv(generator_params=dict(num_loc=100, device=dev), device=dev)
model = RRNet(env, policy=policy, num_augment=8, augment_fn="dihedral8", no_aug_coords=False)
b = case["batch"]
batch = TensorDict({k: v.to(dev) for k, v in b.items()}, batch_size=[64]) if not hasattr(b, "to") else b.to(dev)
for trial in range(int(os.environ.get('DBG_TRIALS', '150'))):
    try:
        out = model.training_step(batch, optimizer=None, world=1, grad_clip=None, seed=case["seed"])
        bad = [n for n, p in policy.named_parameters() if p.grad is not None and not torch.isfinite(p.grad).all()]
        if bad or trial % 25 == 0:
            print(f"trial {trial}: no trace hit; non-finite grads {len(bad)} {bad[:6]}; loss {float(out['loss']):.4f}", flush=True)
        if bad:
            break
    except FloatingPointError as e:
        print(f"trial {trial}:", e, flush=True)
        break
