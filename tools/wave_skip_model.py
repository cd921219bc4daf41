"""Model: how much would a per-wave early exit save in the VRP instance-mode rollout?  Time of a step ~ max over SIMDs of the number of
its tile waves that still have a live rollout (waves w and w+4 share a SIMD; SIMD 3 hosts tile 3 only)."""
import os, sys, torch
ROOT = os.getcwd(); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "real-routing-nco_amd"))
import bench
from rrnco_amd import TensorDict
from rrnco_amd.envs import RCVRPEnv, RMTVRPEnv
from rrnco_amd.models import RRNetPolicy
from rrnco_amd.models.encoder import ATSPInitEmbedding
from rrnco_amd.models.transforms import StateAugmentation
dev = torch.device("cuda")
for name, Env, env_name, B, S, aug, dec in (("C4", RMTVRPEnv, "rcvrptw", 64, 100, True, "multistart_sampling"), ("C3", RCVRPEnv, "rcvrp", 256, 101, False, "multistart_greedy")):
    torch.manual_seed(1234)
    pol = RRNetPolicy(env_name=env_name, embed_dim=128, num_heads=8, num_encoder_layers=6, normalization="instance", use_graph_context=False,
                      nab_type="gating", init_embedding_kwargs=dict(sample_size=25)).to(dev).eval()
    env = Env(generator_params=dict(num_loc=100, device=dev), device=dev) if env_name == "rcvrptw" else Env(generator_params=dict(num_loc=100, device=dev), check_solution=False, device=dev)
    inst = env.generator(B, generator=torch.Generator(device=dev).manual_seed(5))
    td = env.reset(TensorDict(dict(inst.items()), batch_size=[B]))
    if aug:
        td = StateAugmentation(num_augment=8)(td)
    out = pol(td, env, phase="val", decode_type=dec, num_starts=S, return_actions=True, seed=3)
    acts = out["actions"]                                   # [(S * Bp), T] start-major
    Bp = acts.shape[0] // S
    T = acts.shape[1]
    nz = (acts != 0)
    length = T - nz.flip(1).int().argmax(1)                 # index after the last customer visit (+ the closing depot move is not counted)
    length = length.view(S, Bp).t().contiguous()            # [Bp, S]
    pad = torch.zeros(Bp, 112 - S, dtype=length.dtype, device=dev)
    lw = torch.cat([length, pad], 1).view(Bp, 7, 16).max(2).values       # per tile wave: last live step
    wg = lw.max(1).values                                                   # the workgroup's last step
    steps = torch.arange(T, device=dev)[None, None, :]
    live = (steps < lw[:, :, None]).float()                                 # [Bp, 7, T]
    simd = torch.stack([live[:, 0] + live[:, 4], live[:, 1] + live[:, 5], live[:, 2] + live[:, 6], live[:, 3]], 1)   # [Bp, 4, T]
    t_skip = simd.max(1).values.sum(1)                                      # per workgroup: sum over steps of the busiest SIMD's waves
    t_now = 2.0 * wg.float()
    print(name, "T", T, "mean wg steps", float(wg.float().mean()), "live/executed rollout-steps", float(length.float().sum() / (wg.float().sum() * S)),
          "model time with wave skip / now:", float(t_skip.sum() / t_now.sum()))
    # ---- list scheduling of the workgroups on 256 CUs (one resident workgroup each): dispatch order as launched, longest first (oracle),
    # and by a proxy known before the launch (total demand: more routes = more depot returns = more steps)
    import heapq

    def makespan(durs):
        h = [0.0] * 256
        heapq.heapify(h)
        for d in durs:
            heapq.heappush(h, heapq.heappop(h) + d)
        return max(h)
    full = {"C4": 2048, "C3": 512}[name]
    d = wg.float().cpu()
    d = d.repeat((full + len(d) - 1) // len(d))[:full]
    dem = td["demand_linehaul"].sum(-1) if env_name == "rcvrptw" else td["demand"].sum(-1)
    dem = dem.float().cpu().repeat((full + len(dem) - 1) // len(dem))[:full]
    ideal = float(d.sum()) / 256
    print(f"   {full} workgroups: makespan / (sum / 256): as launched {makespan(d.tolist()) / ideal:.3f}, longest first {makespan(d.sort(descending=True).values.tolist()) / ideal:.3f}, "
          f"by total demand {makespan(d[dem.argsort(descending=True)].tolist()) / ideal:.3f}; corr(demand, steps) {float(torch.corrcoef(torch.stack([dem, d]))[0, 1]):.2f}")
