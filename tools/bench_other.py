"""Throughput of the other BASELINE configs (parity-test configs, not bench lines): C3 RCVRP n=100 B=512 POMO S=101,
C4 RCVRPTW n=100 B=256 x8 aug S=100 (sampling).  Prints one JSON line per config."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "real-routing-nco_amd"))
import torch
from rrnco_amd.envs import RCVRPEnv, RMTVRPEnv
from rrnco_amd.models import RRNetPolicy
from rrnco_amd.models.encoder import ATSPInitEmbedding
from rrnco_amd.models.transforms import StateAugmentation
from rrnco_amd.models import rollout as R
from rrnco_amd import TensorDict

dev = torch.device("cuda")
FUSED = os.environ.get("FUSED", "1") != "0"      # FUSED=0: the reference-shaped per-step loop on the step kernels


def policy(env_name):
    torch.manual_seed(1234)
    pol = RRNetPolicy(env_name=env_name, embed_dim=128, num_heads=8, num_encoder_layers=6, normalization="instance",
                      use_graph_context=False, nab_type="gating", init_embedding_kwargs=dict(sample_size=25))
    return pol.to(dev).eval()


def run(name, env, pol, B, S, aug, decode, steps=2):
    inst = env.generator(B, generator=torch.Generator(device=dev).manual_seed(5))
    nd = env.reset(inst)["distance_matrix"]
    sidx = ATSPInitEmbedding.sample_indices(nd, 25)
    if aug:
        sidx = sidx.repeat(8, 1, 1).contiguous()

    def step():
        td = TensorDict(dict(inst.items()), batch_size=[B])
        if aug:
            td = StateAugmentation(augment_fn="dihedral8", no_aug_coords=False)(td)
        td["sample_idx"] = sidx
        return pol(env.reset(td), env, phase="val", decode_type=decode, num_starts=S, seed=1, fused=FUSED)
    out = step(); torch.cuda.synchronize()
    R.TIMING = []
    t0 = time.perf_counter()
    for _ in range(steps):
        out = step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    k = [a.elapsed_time(b) for a, b in R.TIMING] or [0.0]; R.TIMING = None
    print(json.dumps({"config": name, "instances_per_s": B / dt, "ms_per_step": dt * 1e3, "rollout_kernel_ms": sum(k) / len(k),
                      "decode_steps": int(out["actions"].shape[1]), "rollouts": int(out["actions"].shape[0]),
                      "mean_best_cost": float(-out["reward"].view(S, -1).max(0).values.mean())}))


if __name__ == "__main__":
    env = RCVRPEnv(generator_params=dict(num_loc=100, device=dev), check_solution=False, device=dev)
    run("C3 RCVRP n=100 B=512 POMO S=101 greedy", env, policy("rcvrp"), 512, 101, False, "multistart_greedy")
    env = RMTVRPEnv(generator_params=dict(num_loc=100, device=dev), device=dev)
    pol = policy("rcvrptw")
    run("C4 RCVRPTW n=100 B=256 x8 aug S=100 sampling", env, pol, 256, 100, True, "multistart_sampling")
    run("C4' RCVRPTW n=100 B=256 x8 aug S=100 greedy", env, pol, 256, 100, True, "multistart_greedy")
