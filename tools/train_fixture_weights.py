"""Trains the ATSP (or, --problem rcvrp / rcvrptw, a VRP) policy for a short while on the GPU (train.py, the reference's rrnet.yaml hyper-parameters) and writes the
resulting fp32 state_dict as an .npz: the trained weights behind tests/golden/*_trained.npz (oracle/gen_golden.py runs them
through the real reference).  A trained policy has sharper softmaxes and larger activations than default-initialised weights —
the cases the fp16 two-piece arithmetic and the decision-gap contract had not seen (missing #3).
  python tools/train_fixture_weights.py <out.npz> [--problem atsp|rcvrp|rcvrptw] [--steps 1200] [--n 100]"""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "real-routing-nco_amd"))
import numpy as np
import torch
import train

ap = argparse.ArgumentParser()
ap.add_argument("out")
ap.add_argument("--steps", type=int, default=1200)
ap.add_argument("--n", type=int, default=100)
ap.add_argument("--batch", type=int, default=64)
ap.add_argument("--problem", default="atsp", choices=["atsp", "rcvrp", "rcvrptw"])
o = ap.parse_args()
epochs = 4
t0 = time.time()
val = train.main(["--problem", o.problem, "--problem_size", str(o.n), "--epochs", str(epochs), "--batch_size", str(o.batch),
                  "--train_data_size", str(o.steps * o.batch // epochs), "--milestones", "3", "--checkpoint_dir", "/tmp/rr_fixture_ckpt",
                  "--log_every", "100"])
blob = torch.load(f"/tmp/rr_fixture_ckpt/{o.problem}/last.ckpt", map_location="cpu", weights_only=False)
sd = {k[len("policy."):]: v.float().numpy() for k, v in blob["state_dict"].items()}
np.savez_compressed(o.out, **sd)
absmax = max(float(np.abs(v).max()) for v in sd.values())
print(f"trained {o.steps} steps x {o.batch} instances (n = {o.n}) in {time.time() - t0:.0f} s: val/max_aug_reward {val:.4f}; "
      f"{len(sd)} tensors, max |w| {absmax:.3f}, {os.path.getsize(o.out) / 2**20:.1f} MiB -> {o.out}")
