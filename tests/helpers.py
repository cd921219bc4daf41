"""Shared helpers for the parity tests: fixtures, deterministic weights, oracle <-> product glue."""
from __future__ import annotations

import os

import numpy as np
import torch

from oracle import restate

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def fixture_path(name):
    return os.path.join(GOLD, name + ".npz")


def load_fixture(name):
    z = np.load(os.path.join(GOLD, name + ".npz"))
    out = {}
    for k in z.files:
        a = z[k]
        out[k] = torch.from_numpy(a) if a.ndim > 0 else a.item()
    return out


def atsp_weights(fx_or_ss, layers=6, seed=None):
    if isinstance(fx_or_ss, dict) and "weights_file" in fx_or_ss:      # a trained state_dict stored beside the fixtures
        z = np.load(os.path.join(GOLD, str(fx_or_ss["weights_file"])))
        return {k: torch.from_numpy(z[k]).float() for k in z.files}
    if isinstance(fx_or_ss, dict):
        ss, layers, seed = fx_or_ss["sample_size"], fx_or_ss["layers"], fx_or_ss["seed"]
    else:
        ss = fx_or_ss
    t = restate.atsp_weight_template(128, layers, 512, ss)
    if isinstance(fx_or_ss, dict) and fx_or_ss.get("nab_type", "gating") != "gating":
        t = restate.ablation_template(t, fx_or_ss["nab_type"], use_duration=False)
    if isinstance(fx_or_ss, dict) and fx_or_ss.get("normalization", "instance") == "batch":
        t = restate.batchnorm_template(t)
    if isinstance(fx_or_ss, dict) and fx_or_ss.get("normalization", "instance") in ("rms", "layer"):
        t = restate.norm_template(t, fx_or_ss["normalization"])
    if isinstance(fx_or_ss, dict) and "use_coords" in fx_or_ss:      # non-default ATSPInitEmbedding branches (atsp.py:29-35)
        t = restate.atsp_init_variant_template(t, bool(fx_or_ss["use_coords"]), bool(fx_or_ss["use_dist"]))
    return restate.make_weights(t, seed)


def make_policy(w, env_name="atsp", device="cuda"):
    from rrnco_amd.models import RRNetPolicy
    layers = restate.num_layers_of(w)
    ss = [v for k, v in w.items() if k.endswith(".row_embed.weight")][0].shape[1]
    q0 = "encoder.net.layers.0.row_encoding_block.neural_adaptive_bias"
    nab_type = "naive" if (q0 + ".mlp.0.weight") in w else "heuristic" if (q0 + ".alpha") in w else "gating"
    n1 = "encoder.net.layers.0.row_encoding_block.norm1.normalizer"
    norm = ("batch" if (n1 + ".running_mean") in w else "layer" if (n1 + ".weight") not in w
            else "rms" if (n1 + ".bias") not in w else "instance")
    ie = "encoder.init_embedding"
    use_coords = env_name != "atsp" or (ie + ".init_embed.weight") in w
    use_dist = env_name != "atsp" or not use_coords or (ie + ".gating_network_row.gating_fc.0.weight") in w
    pol = RRNetPolicy(env_name=env_name, embed_dim=128, num_heads=8, num_encoder_layers=layers,
                      normalization=norm, use_graph_context=False, nab_type=nab_type,
                      init_embedding_kwargs=dict(use_coords=use_coords, use_polar_feats=True, use_dist=use_dist,
                                                 use_matnet_init=False, sample_type="prob", sample_size=ss))
    pol.load_state_dict(w, strict=True)
    return pol.to(device).eval()


def fixture_state(fx):
    """Raw instance dict (oracle side) of a fixture, incl. augmentation."""
    st = {"locs": fx["locs"], "distance_matrix": fx["distance_matrix"]}
    if fx["aug"]:
        st = restate.augment_state(st)
    return st


def tour_agreement(act_hip, act_ref, gaps=None):
    """fraction of rollouts whose whole tour matches + index of first divergence per rollout (-1 = none)."""
    neq = act_hip != act_ref
    first = torch.where(neq.any(1), neq.float().argmax(1), torch.full((act_ref.shape[0],), -1))
    return float((first < 0).float().mean()), first


def _trained(fx):
    z = np.load(os.path.join(GOLD, str(fx["weights_file"])))
    return {k: torch.from_numpy(z[k]).float() for k in z.files}


def rcvrp_weights(fx):
    if "weights_file" in fx:           # a policy trained on the engine (tools/train_fixture_weights.py --problem rcvrp), run through the reference
        return _trained(fx)
    return restate.make_weights(restate.rcvrp_weight_template(128, fx["layers"], 512, fx["sample_size"]), fx["seed"])


def rcvrp_instance(fx):
    return {k: fx[k] for k in ("locs", "depot", "distance_matrix", "demand")}


def rcvrptw_weights(fx):
    if "weights_file" in fx:
        return _trained(fx)
    t = restate.rcvrptw_weight_template(128, fx["layers"], 512, fx["sample_size"])
    if fx.get("nab_type", "gating") != "gating":
        t = restate.ablation_template(t, fx["nab_type"], use_duration=True)
    return restate.make_weights(t, fx["seed"])


def rcvrptw_instance(fx):
    keys = ("locs", "distance_matrix", "duration_matrix", "demand_linehaul", "time_windows", "service_time")
    opt = ("demand_backhaul", "backhaul_class", "open_route", "distance_limit")            # multi-task variants
    return {k: fx[k] for k in keys + tuple(o for o in opt if o in fx)}


def matnet_weights(fx):
    return restate.make_weights(restate.matnet_weight_template(fx["embed_dim"], fx["heads"], fx["layers"], 512, fx["env_name"]), fx["seed"])


def sampling_law_check(pol, env, inst: dict, sample_idx, S: int, B: int = 2048, seed: int = 11, min_rows: int = 1500):
    """The fused rollout's sampler on B copies of ONE instance: rollouts that share their first action(s) share the distribution of the
    next one, so the empirical frequencies must match exp(reported log-probability) (every category within 5 standard errors), and the
    reported log-probability of a (context, action) pair is one number.  Returns (number of contexts checked, worst deviation in sigmas)."""
    import torch
    from rrnco_amd import TensorDict
    one = {k: v[:1].expand(B, *v.shape[1:]).contiguous().cuda() for k, v in inst.items()}
    td = TensorDict(one, batch_size=[B])
    td["sample_idx"] = sample_idx[:1].expand(B, -1, -1).contiguous().cuda()
    out = pol(env.reset(td), env, phase="val", decode_type="multistart_sampling", num_starts=S, seed=seed, return_actions=True,
              return_sum_log_likelihood=False)
    acts, lps = out["actions"].cpu(), out["log_likelihood"].cpu()
    assert acts.shape[0] == S * B and lps.shape == acts.shape
    NK = int(acts.max()) + 1
    worst, seen = 0.0, 0
    for t in (1, 2):                      # the second action given the start; the third given (start, second)
        ctx = acts[:, 0] if t == 1 else acts[:, 0] * NK + acts[:, 1]
        for c in ctx.unique().tolist():
            rows = ctx == c
            n = int(rows.sum())
            if n < min_rows:
                continue
            a, lp = acts[rows, t], lps[rows, t]
            freq = torch.bincount(a, minlength=NK).double() / n
            p = torch.zeros(NK, dtype=torch.float64)
            p[a] = lp.double().exp()
            for k in a.unique().tolist():                                  # one reported probability per (context, action)
                assert float(lp[a == k].max() - lp[a == k].min()) < 1e-5
            assert (0.98 if t == 1 else 0.95) < float(p.sum()) < 1.0 + 1e-4   # the actions never drawn carry little mass
            sigma = (p * (1 - p) / n).sqrt().clamp_min(1e-3 / n ** 0.5)
            worst = max(worst, float((((freq - p).abs() - 3.0 / n).clamp_min(0) / sigma).max()))      # (3 counts of slack: rare categories are Poisson, not normal)
            seen += 1
    return seen, worst
