"""`-m gpu`: BASELINE.json's FULL sizes through size-independent properties (no oracle run is affordable there):
C2 ATSP n=100 B=512 x8 aug S=100 greedy, C3 RCVRP n=100 B=512 S=101 greedy, C4 RCVRPTW n=100 B=256 x8 aug S=100 sampling.
Every property is recomputed with plain torch ops from the returned action sequences."""
import pytest
import torch

from oracle import restate
from tests import helpers as H

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _policy(env_name, tmpl, seed=1234):
    from rrnco_amd.models import RRNetPolicy
    pol = RRNetPolicy(env_name=env_name, embed_dim=128, num_heads=8, num_encoder_layers=6, normalization="instance",
                      use_graph_context=False, nab_type="gating", init_embedding_kwargs=dict(sample_size=25))
    pol.load_state_dict(restate.make_weights(tmpl, seed), strict=True)
    return pol.to(DEV).eval()


def _route_cost(D, acts, b_of_r, closed_by_depot):
    """sum_t D[b, a_t, a_t+1] recomputed with gathers; VRP routes start and end at the depot (node 0)."""
    if closed_by_depot:
        z = torch.zeros(acts.shape[0], 1, dtype=acts.dtype, device=acts.device)
        path = torch.cat([z, acts, z], 1)
        frm, to = path[:, :-1], path[:, 1:]
    else:
        frm, to = acts, acts.roll(-1, 1)
    return D[b_of_r[:, None], frm, to].double().sum(1)


def test_c2_atsp_full_batch_properties():
    from rrnco_amd.envs import ATSPEnv, ATSPGenerator
    from rrnco_amd.models.encoder import ATSPInitEmbedding
    from rrnco_amd.models.transforms import StateAugmentation
    from rrnco_amd.ops import unbatchify
    B, N, S = 512, 100, 100
    pol = _policy("atsp", restate.atsp_weight_template())
    env = ATSPEnv(generator_params=dict(num_loc=N, device=DEV), check_solution=False, device=DEV)
    inst = ATSPGenerator(num_loc=N, device=DEV)(B, generator=torch.Generator(device=DEV).manual_seed(11))
    td = env.reset(StateAugmentation(augment_fn="dihedral8", no_aug_coords=False)(inst))
    td["sample_idx"] = ATSPInitEmbedding.sample_indices(td["distance_matrix"], 25)
    out = pol(td, env, phase="val", decode_type="multistart_greedy", num_starts=S)
    acts = out["actions"]
    R = S * 8 * B
    assert acts.shape == (R, N)
    assert bool((acts.sort(1).values == torch.arange(N, device=DEV)).all())                          # atsp/env.py:213-220
    assert torch.equal(acts[:, 0], torch.arange(S, device=DEV).repeat_interleave(8 * B))             # POMO start nodes
    b_of_r = torch.arange(R, device=DEV) % (8 * B)
    cost = _route_cost(td["distance_matrix"], acts, b_of_r, closed_by_depot=False)
    assert float((out["normalized_reward"].double() + cost).abs().max()) < 2e-4                     # atsp/env.py:192-211
    real = out["normalized_reward"] * (td["max_distance"] - td["min_distance"] + 1e-6)[b_of_r] + 0   # de-normalised edge sum
    real = real + N * td["min_distance"][b_of_r]
    assert float((out["reward"] - real).abs().max() / out["reward"].abs().max()) < 1e-5
    assert bool(torch.isfinite(out["log_likelihood"]).all()) and bool((out["log_likelihood"] <= 0).all())
    rew = unbatchify(out["reward"], (8, S))
    assert rew.shape == (B, 8, S) and bool((rew.amax(dim=(1, 2)) >= rew[:, 0].amax(-1)).all())


def test_c3_rcvrp_full_batch_feasibility():
    from rrnco_amd.envs import RCVRPEnv
    from rrnco_amd.models.encoder import ATSPInitEmbedding
    B, N, S = 512, 100, 101
    pol = _policy("rcvrp", restate.rcvrp_weight_template())
    env = RCVRPEnv(generator_params=dict(num_loc=N, device=DEV), check_solution=False, device=DEV)
    inst = env.generator(B, generator=torch.Generator(device=DEV).manual_seed(12))
    td = env.reset(inst)
    td["sample_idx"] = ATSPInitEmbedding.sample_indices(td["distance_matrix"], 25)
    out = pol(td, env, phase="val", decode_type="multistart_greedy", num_starts=S)
    acts = out["actions"]
    R, T = acts.shape
    assert R == S * B
    b_of_r = torch.arange(R, device=DEV) % B
    # every customer exactly once (rcvrp/env.py:236-249)
    cnt = torch.zeros(R, N + 1, device=DEV).scatter_add_(1, acts, torch.ones(R, T, device=DEV))
    assert bool((cnt[:, 1:] == 1).all())
    # capacity per route: demand accumulated between depot visits never exceeds the (normalised) capacity 1
    dem = torch.cat([torch.zeros(B, 1, device=DEV), td["demand"]], 1)[b_of_r[:, None], acts]
    seg = (acts == 0).cumsum(1)
    load = torch.zeros(R, T + 1, device=DEV).scatter_add_(1, seg, dem)
    assert float(load.max()) <= 1.0 + 1e-5
    cost = _route_cost(td["distance_matrix"], acts, b_of_r, closed_by_depot=True)
    assert float((out["normalized_reward"].double() + cost).abs().max()) < 5e-4                     # rcvrp/env.py:197-219
    out2 = pol(td, env, phase="val", decode_type="multistart_greedy", num_starts=S)                  # bit-identical repeat
    assert torch.equal(out2["actions"], acts) and torch.equal(out2["log_likelihood"], out["log_likelihood"])


def test_c4_rcvrptw_full_batch_sampling_feasibility():
    from rrnco_amd.envs import RMTVRPEnv
    from rrnco_amd.models.encoder import ATSPInitEmbedding
    from rrnco_amd.models.transforms import StateAugmentation
    B, N, S = 256, 100, 100
    pol = _policy("rcvrptw", restate.rcvrptw_weight_template())
    env = RMTVRPEnv(generator_params=dict(num_loc=N, device=DEV), device=DEV)
    inst = env.generator(B, generator=torch.Generator(device=DEV).manual_seed(13))
    td = env.reset(StateAugmentation(augment_fn="dihedral8", no_aug_coords=False)(inst))
    Bp = 8 * B
    td["sample_idx"] = ATSPInitEmbedding.sample_indices(td["distance_matrix"], 25)
    D, Dur = td["distance_matrix"], td["duration_matrix"]
    tw, service, demand = td["time_windows"], td["service_time"], td["demand_linehaul"]
    out = pol(td, env, phase="val", decode_type="multistart_sampling", num_starts=S, seed=3)
    acts = out["actions"]
    R, T = acts.shape
    assert R == S * Bp
    b_of_r = torch.arange(R, device=DEV) % Bp
    cnt = torch.zeros(R, N + 1, device=DEV).scatter_add_(1, acts, torch.ones(R, T, device=DEV))
    assert bool((cnt[:, 1:] == 1).all())
    # replay rmtvrp/env.py:155-215: time, load; every arrival inside its window, every return to the depot before it closes
    t = torch.zeros(R, device=DEV); load = torch.zeros(R, device=DEV); prev = torch.zeros(R, dtype=torch.long, device=DEV)
    worst_late, worst_load = -1e9, 0.0
    for k in range(T):
        a = acts[:, k]
        arrival = t + Dur[b_of_r, prev, a]
        worst_late = max(worst_late, float((arrival - tw[b_of_r, a, 1]).max()))
        nz = (a != 0).float()
        t = nz * (torch.maximum(arrival, tw[b_of_r, a, 0]) + service[b_of_r, a])
        load = nz * (load + demand[b_of_r, a])
        worst_load = max(worst_load, float(load.max()))
        prev = a
    assert worst_late < 1e-5 and worst_load <= 1.0 + 1e-5
    cost = _route_cost(D, acts, b_of_r, closed_by_depot=True)
    assert float((out["normalized_reward"].double() + cost).abs().max()) < 1e-3
    ll = out["log_likelihood"]
    assert bool(torch.isfinite(ll).all()) and bool((ll <= 0).all())
    out2 = pol(td, env, phase="val", decode_type="multistart_sampling", num_starts=S, seed=3)        # counter-based generator
    assert torch.equal(out2["actions"], acts) and torch.equal(out2["log_likelihood"], ll)              # bit-identical repeat
    out3 = pol(td, env, phase="val", decode_type="multistart_sampling", num_starts=S, seed=4)
    assert not torch.equal(out3["actions"][:, :out2["actions"].shape[1]][: , :10], acts[:, :10])


def test_c2_split_bf16_pointer_mlp_rollout_equals_the_fp32_mfma_rollout(monkeypatch):
    """The default rollout runs the pointer MLP on the bf16 matrix pipe with 3-way split fp32 operands.  At BASELINE configs[1]
    size (409 600 rollouts, 40 M decisions) its greedy tours must equal the fp32-MFMA build's on >= 99.9 % of the rollouts, and
    wherever they part the fp32 build itself must see a near tie: the fp32 log-probability of the node the split build chose
    (teacher-forced through the fp32 build, same prefix) is within 1e-3 of the fp32 maximum."""
    from rrnco_amd.envs import ATSPEnv, ATSPGenerator
    from rrnco_amd.models import rollout as R
    from rrnco_amd.models.encoder import ATSPInitEmbedding
    from rrnco_amd.models.transforms import StateAugmentation
    B, N, S = 512, 100, 100
    pol = _policy("atsp", restate.atsp_weight_template())
    env = ATSPEnv(generator_params=dict(num_loc=N, device=DEV), check_solution=False, device=DEV)
    inst = ATSPGenerator(num_loc=N, device=DEV)(B, generator=torch.Generator(device=DEV).manual_seed(12))
    td = env.reset(StateAugmentation(augment_fn="dihedral8", no_aug_coords=False)(inst))
    td["sample_idx"] = ATSPInitEmbedding.sample_indices(td["distance_matrix"], 25)
    kw = dict(phase="val", decode_type="multistart_greedy", num_starts=S, return_sum_log_likelihood=False)
    monkeypatch.setattr(R, "SPLIT_MLP", False)        # (the encoder is the same build in both runs: same embeddings)
    a32 = pol(td.clone(), env, **kw)
    monkeypatch.setattr(R, "SPLIT_MLP", True)
    asp = pol(td.clone(), env, **kw)
    neq = a32["actions"] != asp["actions"]
    differs = neq.any(1)
    frac = 1.0 - float(differs.float().mean())
    best32 = a32["reward"].view(S, 8 * B).amax(0).view(8, B).amax(0)
    bestsp = asp["reward"].view(S, 8 * B).amax(0).view(8, B).amax(0)
    print(f"\n[C2] tours equal on {100 * frac:.4f} % of 409 600 rollouts; instances with the identical best cost: "
          f"{100 * float((best32 == bestsp).float().mean()):.2f} %")
    assert frac >= 0.999
    if bool(differs.any()):
        monkeypatch.setattr(R, "SPLIT_MLP", False)
        ev = pol(td.clone(), env, phase="val", actions=asp["actions"][:, 1:], num_starts=S, return_sum_log_likelihood=False)
        rows = torch.nonzero(differs).flatten()
        t = neq[rows].float().argmax(1)                                  # first step where the tours part
        lp_max = a32["log_likelihood"][rows, t]                          # fp32 log-probability of its own (greedy = max) choice
        lp_other = ev["log_likelihood"][rows, t]                         # ... of the split build's choice, same prefix
        gap = lp_max - lp_other
        print(f"[C2] {rows.numel()} diverging rollouts, largest fp32 decision gap at the parting step {float(gap.max()):.2e}")
        assert float(gap.min()) > -1e-5 and float(gap.max()) < 1e-3


def test_c5_training_step_full_size_properties():
    """BASELINE configs[4] per-GPU shard (ATSP n=100, 512 instances, 100 sampled starts): sampled tours are permutations, the
    backward's replayed log-likelihood is the rollout's, every gradient is finite."""
    from rrnco_amd.envs import ATSPEnv, ATSPGenerator
    from rrnco_amd.models.rl import RRNet
    B, N, S = 512, 100, 100
    pol = _policy("atsp", restate.atsp_weight_template()).train()
    env = ATSPEnv(generator_params=dict(num_loc=N, device=DEV), check_solution=False, device=DEV)
    model = RRNet(env, policy=pol)
    batch = ATSPGenerator(num_loc=N, device=DEV)(B, generator=torch.Generator(device=DEV).manual_seed(21))
    out = model.training_step(batch, seed=3, grad_clip=1.0)
    acts = out["actions"]
    assert acts.shape == (S * B, N) and bool((acts.sort(1).values == torch.arange(N, device=DEV)).all())
    assert torch.allclose(out["replay_log_likelihood"], out["log_likelihood"], rtol=2e-5, atol=2e-3)
    assert bool(torch.isfinite(out["log_likelihood"]).all()) and float(out["grad_norm"]) > 0
    total = torch.linalg.vector_norm(torch.stack([p.grad.norm() for p in pol.parameters()]))
    assert bool(torch.isfinite(total)) and float(total) <= 1.0 + 1e-4                  # clipped to gradient_clip_val = 1.0
