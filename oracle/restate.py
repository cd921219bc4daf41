"""ORACLE — CPU restatement of the RRNCO construction-rollout hot path (TEST INFRASTRUCTURE).

Plain torch-CPU fp32, op-for-op with the reference so that, on the same machine and the same
inputs, it reproduces the reference bit-for-bit (checked by `oracle/gen_golden.py`, which runs the
real reference through `oracle/ref_shim.py` in the build container and refuses to write a fixture
unless this file agrees exactly).  Pinned against: the committed `tests/golden/*.npz` fixtures
(reference outputs).  The reference's own test-suite holds no numeric vectors for this path
(SURVEY.md §4), so those fixtures are the pin.

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import this
module.  The product (`rrnco_amd`) never does: it fails loudly if the HIP library is missing.

Weights are a flat ``dict[str, Tensor]`` using the reference's ``state_dict`` names
(`RRNetPolicy.state_dict()`), state is a plain ``dict[str, Tensor]``.

Reference citations are relative to /root/reference.
"""
from __future__ import annotations

import math
from typing import Dict, Optional

import torch
import torch.nn.functional as F

Tensor = torch.Tensor
W = Dict[str, Tensor]


# ----------------------------------------------------------------------------------------------
# rl4co ops (rl4co 0.6.0 `rl4co/utils/ops.py`, absent from /root/reference; call sites:
# rrnco/models/decoding.py:189,203  rrnco/models/decoder.py:173,187  rrnco/models/rl.py:112)
# ----------------------------------------------------------------------------------------------
def batchify(x: Tensor, n: int) -> Tensor:
    """repeat-major: out[r*B + b] = x[b]."""
    if n <= 0:
        return x
    s = x.shape
    return x.expand(n, *s).contiguous().view(s[0] * n, *s[1:])


def unbatchify(x: Tensor, n: int) -> Tensor:
    """'(r b) ... -> b r ...'."""
    if n <= 0:
        return x
    s = x.shape
    return x.view(n, s[0] // n, *s[1:]).permute(1, 0, *range(2, len(s) + 1))


def batchify_state(td: dict, n: int) -> dict:
    return {k: batchify(v, n) for k, v in td.items()}


def gather_by_index(src: Tensor, idx: Tensor, dim: int = 1, squeeze: bool = True) -> Tensor:
    expanded_shape = list(src.shape)
    expanded_shape[dim] = -1
    idx = idx.view(idx.shape + (1,) * (src.dim() - idx.dim())).expand(expanded_shape)
    squeeze = idx.size(dim) == 1 and squeeze
    return src.gather(dim, idx).squeeze(dim) if squeeze else src.gather(dim, idx)


def lin(w: W, name: str, x: Tensor) -> Tensor:
    return F.linear(x, w[name + ".weight"], w.get(name + ".bias"))


# ----------------------------------------------------------------------------------------------
# Synthetic instances (rrnco/envs/atsp/generator_lazy.py:208-237)
# ----------------------------------------------------------------------------------------------
def atsp_synthetic(batch: int, n: int, seed: int) -> dict:
    g = torch.Generator().manual_seed(seed)
    locs = torch.rand(batch, n, 2, generator=g)
    dms = torch.rand(batch, n, n, generator=g)
    dms[..., torch.arange(n), torch.arange(n)] = 0
    for i in range(n):  # TMAT class: triangle closure
        dms = torch.minimum(dms, dms[..., :, [i]] + dms[..., [i], :])
    return {"locs": locs, "distance_matrix": dms}


def dihedral8(xy: Tensor) -> Tensor:
    """rrnco/models/utils/transforms.py:15-37 — 8 blocks stacked on dim 0 (aug-major)."""
    x, y = xy.split(1, dim=2)
    zs = [(x, y), (1 - x, y), (x, 1 - y), (1 - x, 1 - y), (y, x), (1 - y, x), (y, 1 - x), (1 - y, 1 - x)]
    return torch.cat([torch.cat(z, dim=2) for z in zs], dim=0)


def augment_state(td: dict, num_augment: int = 8) -> dict:
    """StateAugmentation(augment_fn='dihedral8', no_aug_coords=False) as test.py:28 builds it
    (transforms.py:142-154): batchify everything x8, replace `locs` by the 8 reflections of the
    first 1/8 (transforms.py:40-47)."""
    assert num_augment == 8
    out = batchify_state(td, 8)
    b = td["locs"].shape[0]
    out["locs"] = dihedral8(out["locs"][:b])
    return out


# ----------------------------------------------------------------------------------------------
# ATSP env (rrnco/envs/atsp/env.py)
# ----------------------------------------------------------------------------------------------
def atsp_reset(td: dict, normalize: bool = True) -> dict:
    """env.py:107-155 (+ RL4COEnvBase.reset adds done=False)."""
    distance = td["distance_matrix"]
    B = distance.shape[0]
    out = {}
    if normalize:
        mn = distance.amin(dim=(-2, -1), keepdim=True)
        mx = distance.amax(dim=(-2, -1), keepdim=True)
        distance = ((distance - mn) / (mx - mn + 1e-6)).to(torch.float32)
        out["min_distance"] = mn.squeeze(-1).squeeze(-1)
        out["max_distance"] = mx.squeeze(-1).squeeze(-1)
    n = distance.shape[-1]
    out.update(
        distance_matrix=distance,
        first_node=torch.zeros(B, 1, dtype=torch.int64),
        current_node=torch.zeros(B, 1, dtype=torch.int64),
        i=torch.zeros(B, 1, dtype=torch.int64),
        action_mask=torch.ones(B, n, dtype=torch.bool),
        done=torch.zeros(B, 1, dtype=torch.bool),
    )
    if "locs" in td:
        out["locs"] = td["locs"]
    return out


def atsp_step(td: dict) -> dict:
    """env.py:80-105."""
    cur = td["action"]
    first = cur if td["i"].flatten()[0].item() == 0 else td["first_node"]
    avail = td["action_mask"].scatter(-1, cur.unsqueeze(-1).expand_as(td["action_mask"]), 0)
    done = torch.count_nonzero(avail, dim=-1) <= 0
    td.update(first_node=first, current_node=cur, i=td["i"] + 1, action_mask=avail,
              reward=torch.zeros_like(done), done=done)
    return td


def atsp_reward(td: dict, actions: Tensor, normalize: bool = True):
    """env.py:192-211 — returns (real, normalized) when normalize."""
    D = td["distance_matrix"]
    src, tgt = actions, torch.roll(actions, -1, dims=1)
    bidx = torch.arange(D.shape[0]).unsqueeze(1)
    if normalize:
        nd = -D[bidx, src, tgt].sum(-1)
        real = nd * (td["max_distance"] - td["min_distance"] + 1e-6) + td["min_distance"]
        return real, nd
    return -D[bidx, src, tgt].sum(-1)


def atsp_check(actions: Tensor) -> bool:
    """env.py:213-220."""
    n = actions.size(1)
    return bool((torch.arange(n).view(1, -1).expand_as(actions) == actions.sort(1)[0]).all())


# ----------------------------------------------------------------------------------------------
# Init embedding (rrnco/models/env_embeddings/atsp.py)
# ----------------------------------------------------------------------------------------------
def sample_neighbor_indices(distance: Tensor, sample_size: int, generator=None) -> Tensor:
    """atsp.py:55-67 ('prob' sampling).  The index tensor is an explicit input to both the
    oracle and the HIP path: this helper only produces one."""
    B, N, _ = distance.shape
    idx = torch.arange(N)
    pd = distance.clone()
    pd[:, idx, idx] = 1e6
    inv = 1 / (pd + 1e-6)
    prob = (inv / inv.sum(dim=-1, keepdim=True)).reshape(B * N, -1)
    return torch.multinomial(prob, sample_size, replacement=False, generator=generator).reshape(B, N, sample_size)


def contextual_gating(w: W, p: str, coord: Tensor, dist: Tensor) -> Tensor:
    """atsp.py:108-121 — scalar gate per node."""
    comb = torch.cat([coord, dist], dim=-1)
    g = torch.sigmoid(lin(w, p + ".gating_fc.2", F.relu(lin(w, p + ".gating_fc.0", comb))))
    return g * coord + (1 - g) * dist


def atsp_init_embedding(w: W, locs: Tensor, distance: Tensor, sidx: Tensor):
    """atsp.py:69-104.  The branch follows from the parameters the state_dict holds (atsp.py:29-35): init_embed exists with use_coords,
    the gates with use_coords and use_dist.  The index tensor is an input whatever law drew it (sample_type "prob" / "random")."""
    p = "encoder.init_embedding"
    if (p + ".init_embed.weight") not in w:            # use_coords=False (:94-104): unsorted gathers straight into row_embed / col_embed
        return (lin(w, p + ".row_embed", distance.gather(2, sidx)), lin(w, p + ".col_embed", distance.transpose(1, 2).gather(2, sidx)))
    if (p + ".gating_network_row.gating_fc.0.weight") not in w:      # use_coords, not use_dist (:92)
        node = lin(w, p + ".init_embed", locs.to(w[p + ".init_embed.weight"].dtype))
        return node.clone(), node.clone()
    node = lin(w, p + ".init_embed", locs.to(w[p + ".init_embed.weight"].dtype))      # (.float() in the reference: fp32 weights; a float64 run of the oracle keeps its dtype)
    rowd = distance.gather(2, sidx)
    cold = distance.transpose(1, 2).gather(2, sidx)
    row = lin(w, p + ".row_embed", rowd.sort(dim=-1).values)
    col = lin(w, p + ".col_embed", cold.sort(dim=-1).values)
    return (contextual_gating(w, p + ".gating_network_row", node, row),
            contextual_gating(w, p + ".gating_network_col", node, col))


# ----------------------------------------------------------------------------------------------
# Encoder net (rrnco/models/nn/attn_freenet.py)
# ----------------------------------------------------------------------------------------------
def instance_norm(w: W, p: str, x: Tensor) -> Tensor:
    """Normalization attn_freenet.py:101-105: InstanceNorm1d(E, affine) over nodes, or — when the weights carry running
    statistics (normalization='batch', the constructor default) — BatchNorm1d in eval mode over the flattened B*N rows."""
    if (p + ".normalizer.running_mean") in w:
        return F.batch_norm(x.reshape(-1, x.size(-1)), w[p + ".normalizer.running_mean"], w[p + ".normalizer.running_var"],
                            weight=w[p + ".normalizer.weight"], bias=w[p + ".normalizer.bias"], training=False,
                            eps=1e-5).view(*x.size())
    if (p + ".normalizer.weight") not in w:          # normalization='layer' (:106-109): no parameters, torch.var is unbiased
        return (x - x.mean((1, 2)).view(-1, 1, 1)) / torch.sqrt(x.var((1, 2)).view(-1, 1, 1) + 1e-05)
    if (p + ".normalizer.bias") not in w:            # normalization='rms': RMSNorm (:13-26, 110-111), weight only
        return x * torch.rsqrt(x.pow(2).mean(-1, keepdim=True) + 1e-5) * w[p + ".normalizer.weight"]
    return F.instance_norm(x.permute(0, 2, 1), weight=w[p + ".normalizer.weight"],
                           bias=w[p + ".normalizer.bias"], eps=1e-5).permute(0, 2, 1)


def atsp_init_variant_template(template: Dict[str, tuple], use_coords: bool, use_dist: bool) -> Dict[str, tuple]:
    """state_dict template of the ATSP policy built with use_coords / use_dist switched off (atsp.py:29-35)."""
    p = "encoder.init_embedding"
    t = dict(template)
    if not use_coords:
        t = {k: v for k, v in t.items() if not k.startswith(p + ".init_embed.")}
    if not (use_coords and use_dist):
        t = {k: v for k, v in t.items() if not k.startswith(p + ".gating_network_")}
    return t


def norm_template(template: Dict[str, tuple], normalization: str) -> Dict[str, tuple]:
    """state_dict template of the same policy built with normalization='rms' (weight only) or 'layer' (no parameters)."""
    if normalization == "rms":
        return {k: v for k, v in template.items() if not k.endswith(".normalizer.bias")}
    if normalization == "layer":
        return {k: v for k, v in template.items() if ".normalizer." not in k}
    return template


def batchnorm_template(template: Dict[str, tuple]) -> Dict[str, tuple]:
    """state_dict template of the same policy built with normalization='batch': BatchNorm1d buffers next to every affine."""
    t = {}
    for k, v in template.items():
        t[k] = v
        if k.endswith(".normalizer.bias"):
            base = k[: -len("bias")]
            t[base + "running_mean"] = v; t[base + "running_var"] = v; t[base + "num_batches_tracked"] = ()
    return t


def pairwise_angles(coords: Tensor) -> Tensor:
    """attn_freenet.py:254-262."""
    d = coords.unsqueeze(2) - coords.unsqueeze(1)
    return torch.atan2(d[..., 1], d[..., 0])


def nab_gating(w: W, p: str, coords: Tensor, cost: Tensor, dur: Optional[Tensor]) -> Tensor:
    """DistAngleFusion.forward attn_freenet.py:242-289."""
    def mlp(q, x):
        return lin(w, q + ".2", F.relu(lin(w, q + ".0", x.unsqueeze(-1))))
    angles = pairwise_angles(coords)
    de = mlp(p + ".dist_emb", cost)
    ae = mlp(p + ".angle_emb", angles)
    if dur is not None:
        du = mlp(p + ".dur_emb", dur)
        gi = torch.cat([de, ae, du], dim=-1)
        logits = lin(w, p + ".gate.2", F.silu(lin(w, p + ".gate.0", gi)))
        g = F.softmax(logits / w[p + ".gate_temperature"].exp(), dim=-1)
        fused = g[..., [0]] * de + g[..., [1]] * ae + g[..., [2]] * du
    else:
        g = torch.sigmoid(lin(w, p + ".gate.0", torch.cat([de, ae], dim=-1)))
        fused = g * de + (1 - g) * ae
    return lin(w, p + ".out_lin", fused).squeeze(-1)


def aft_full(w: W, p: str, x: Tensor, y: Tensor, bias: Tensor) -> Tensor:
    """AFTFull.forward attn_freenet.py:309-327 (note exp(softmax(.)) twice)."""
    Q = lin(w, p + ".to_q", x)
    K = lin(w, p + ".to_k", y)
    V = lin(w, p + ".to_v", y)
    a = torch.softmax(bias, dim=-1)
    K = torch.softmax(K, dim=1)
    temp = torch.exp(a) @ torch.mul(torch.exp(K), V)
    weighted = temp / (torch.exp(a) @ torch.exp(K))
    return lin(w, p + ".project", torch.mul(torch.sigmoid(Q), weighted))


def nab_heuristic(w: W, p: str, cost: Tensor, dur: Optional[Tensor]) -> Tensor:
    """HeuristicNeuralAdaptiveBias.forward attn_freenet.py:138-167: -log2(N) * d_ij (the module's own `alpha` is unused)."""
    log2n = torch.log2(torch.tensor(cost.size(-1), dtype=cost.dtype))
    if dur is not None and (p + ".distance_weight") in w:
        d = w[p + ".distance_weight"] * cost + w[p + ".duration_weight"] * dur
    else:
        d = cost
    return -log2n * d


def nab_naive(w: W, p: str, coords: Tensor, cost: Tensor, dur: Optional[Tensor]) -> Tensor:
    """NaiveNeuralAdaptiveBias.forward attn_freenet.py:182-199 (needs the duration matrix: torch.cat fails on None)."""
    if dur is None:
        raise TypeError("NaiveNeuralAdaptiveBias concatenates duration_mat; the reference raises without it")
    x = torch.stack([pairwise_angles(coords), cost, dur], dim=-1)
    return lin(w, p + ".mlp.2", F.silu(lin(w, p + ".mlp.0", x))).squeeze(-1)


def nab_any(w: W, p: str, nab_name: str, coords: Tensor, cost: Tensor, dur) -> Tensor:
    """AttnFree_Block's choice of bias module (attn_freenet.py:379-405), recognised by the parameters present."""
    q = p + ".neural_adaptive_bias"
    if (q + ".mlp.0.weight") in w:
        return nab_naive(w, q, coords, cost, dur)
    if (q + ".alpha") in w:
        return nab_heuristic(w, q, cost, dur)
    return nab_gating(w, p + "." + nab_name, coords, cost, dur)


def ablation_template(template: Dict[str, tuple], nab_type: str, use_duration: bool, embed_dim: int = 128) -> Dict[str, tuple]:
    """state_dict template of the same policy built with nab_type in {'naive', 'heuristic'} (configs/experiment/
    rrnet_naive.yaml, rrnet_heuristic.yaml): the gating fusion's parameters are replaced by the ablation module's."""
    t = {k: v for k, v in template.items() if ".angle_distance_fusion." not in k and ".neural_adaptive_bias." not in k}
    blocks = sorted({k.rsplit(".alpha", 1)[0] for k in template if k.endswith("_encoding_block.alpha")})
    for b in blocks:
        q = b + ".neural_adaptive_bias"
        if nab_type == "naive":
            c = 3 if use_duration else 2
            t[q + ".mlp.0.weight"] = (embed_dim, c); t[q + ".mlp.0.bias"] = (embed_dim,)
            t[q + ".mlp.2.weight"] = (1, embed_dim); t[q + ".mlp.2.bias"] = (1,)
        elif nab_type == "heuristic":
            t[q + ".alpha"] = (1,)
            if use_duration:
                t[q + ".distance_weight"] = (1,); t[q + ".duration_weight"] = (1,)
        else:
            raise ValueError(nab_type)
    return t


def block(w: W, p: str, row: Tensor, col: Tensor, cost: Tensor, coords: Tensor, dur, nab_name: str) -> Tensor:
    """AttnFree_Block.forward attn_freenet.py:417-441."""
    row = instance_norm(w, p + ".norm1", row)
    col = instance_norm(w, p + ".norm2", col)
    bias = nab_any(w, p, nab_name, coords, cost, dur) * w[p + ".alpha"]
    out = aft_full(w, p + ".attn_free", row, col, bias)
    out = instance_norm(w, p + ".norm3", lin(w, p + ".multi_head_combine", out))
    f = p + ".feed_forward.ops"
    x = instance_norm(w, f + ".norm1", row + out)
    return instance_norm(w, f + ".norm2", x + lin(w, f + ".ffn.W2", F.relu(lin(w, f + ".ffn.W1", x))))


def encoder_net(w: W, row: Tensor, col: Tensor, cost: Tensor, coords: Tensor, dur=None, num_layers: int = 6):
    """AttnFreeNet / Attn_Free_Layer attn_freenet.py:472-488, 517-521."""
    nab = "neural_adaptive_bias" if dur is not None else "angle_distance_fusion"
    for l in range(num_layers):
        p = f"encoder.net.layers.{l}"
        r = block(w, p + ".row_encoding_block", row, col, cost, coords, dur, nab)
        c = block(w, p + ".col_encoding_block", col, row, cost.transpose(1, 2), coords,
                  None if dur is None else dur.transpose(1, 2), nab)
        row, col = r, c
    return row, col


def num_layers_of(w: W) -> int:
    return 1 + max(int(k.split(".")[3]) for k in w if k.startswith("encoder.net.layers."))


def atsp_encoder(w: W, td: dict, sidx: Tensor):
    """RRNetEncoder.forward encoder.py:80-112 for env_name='atsp'."""
    row, col = atsp_init_embedding(w, td["locs"], td["distance_matrix"], sidx)
    return encoder_net(w, row, col, td["distance_matrix"], td["locs"].to(row.dtype), None, num_layers_of(w))


# ----------------------------------------------------------------------------------------------
# Decoder (rrnco/models/decoder.py) + decoding (rrnco/models/decoding.py)
# ----------------------------------------------------------------------------------------------
def precompute_cache(w: W, row_emb: Tensor, col_emb: Tensor) -> dict:
    """decoder.py:214-232."""
    gk, gv, lk = F.linear(col_emb, w["decoder.project_node_embeddings.weight"]).chunk(3, dim=-1)
    return dict(node_embeddings=row_emb, glimpse_key=gk, glimpse_val=gv, logit_key=lk)


def atsp_context(w: W, emb: Tensor, td: dict) -> Tensor:
    """rl4co TSPContext (absent from the tree).  td is [B,S]-shaped here."""
    B = emb.size(0)
    if td["i"].flatten()[0].item() < 1:
        ph = w["decoder.context_embedding.W_placeholder"]
        if td["_two_d"]:
            ctx = ph[None, None, :].expand(B, td["current_node"].shape[1], ph.size(-1))
        else:
            ctx = ph[None, :].expand(B, ph.size(-1))
    else:
        node_dim = (-1,) if td["first_node"].dim() == 1 else (td["first_node"].size(-1), -1)
        ctx = gather_by_index(emb, torch.stack([td["first_node"], td["current_node"]], -1).view(B, -1)).view(B, *node_dim)
    return F.linear(ctx, w["decoder.context_embedding.project_context.weight"])


def pointer(w: W, q: Tensor, k: Tensor, v: Tensor, lk: Tensor, mask: Tensor, num_heads: int = 8) -> Tensor:
    """RRNet_PointerAttention.forward decoder.py:281-323."""
    def heads(t):  # '... g (h s) -> ... h g s'
        return t.unflatten(-1, (num_heads, -1)).transpose(-2, -3)
    am = mask.unsqueeze(1) if mask.ndim == 3 else mask.unsqueeze(1).unsqueeze(2)
    h = F.scaled_dot_product_attention(heads(q), heads(k), heads(v), attn_mask=am)
    h = h.transpose(-2, -3).flatten(-2)
    g = h + q
    g = F.linear(F.relu(F.linear(g, w["decoder.pointer.ffn.lins.0.weight"], w["decoder.pointer.ffn.lins.0.bias"])),
                 w["decoder.pointer.ffn.lins.1.weight"], w["decoder.pointer.ffn.lins.1.bias"]) + g
    logits = torch.bmm(g, lk.squeeze(-2).transpose(-2, -1)).squeeze(-2) / math.sqrt(g.size(-1))
    return logits


def atsp_decoder_step(w: W, td_flat: dict, cache: dict, S: int):
    """RRNetDecoder.forward decoder.py:151-206 for atsp (multistart: unbatchify to [B,S,..])."""
    if S > 1:
        td = {k: unbatchify(v, S) for k, v in td_flat.items() if k in ("first_node", "current_node", "i", "action_mask")}
        td["_two_d"] = True
        D = cache["_D"]  # [B,N,N] un-batchified (== unbatchify(td["distance_matrix"])[:,0])
    else:
        td = {k: td_flat[k] for k in ("first_node", "current_node", "i", "action_mask")}
        td["_two_d"] = False
        D = td_flat["distance_matrix"]
    q = atsp_context(w, cache["node_embeddings"], td)
    q = q.unsqueeze(1) if q.ndim == 2 else q
    mask = td["action_mask"]
    logits = pointer(w, q, cache["glimpse_key"], cache["glimpse_val"], cache["logit_key"], mask)
    if S > 1:
        # reference gathers from the S-times batchified matrix; identical values
        bias = w["decoder.alpha"] * gather_by_index(D.unsqueeze(1).expand(-1, S, -1, -1), td["current_node"], dim=-2)
    else:
        bias = w["decoder.alpha"] * gather_by_index(D, td["current_node"], dim=-2)
    _ft = torch.float64 if logits.dtype == torch.float64 else torch.float32      # (decoder.py:195-196 casts to fp32; a float64 run of the oracle stays float64)
    logits = torch.log(torch.exp(logits.to(_ft) - bias.to(_ft)) + 1e-6)
    if S > 1:
        logits = logits.permute(1, 0, 2).reshape(-1, logits.shape[-1])  # 'b s l -> (s b) l'
        mask = mask.permute(1, 0, 2).reshape(-1, mask.shape[-1])
    return logits, mask


def process_logits(logits: Tensor, mask: Tensor, temperature: float = 1.0, tanh_clipping: float = 10.0,
                   top_p: float = 0.0, top_k: int = 0) -> Tensor:
    """decoding.py:311-361, with the top-k (:37-42) and top-p (:45-63) filters."""
    if tanh_clipping > 0:
        logits = torch.tanh(logits) * tanh_clipping
    logits = logits.clone()
    logits[~mask] = float("-inf")
    logits = logits / temperature
    if top_k > 0:
        k = min(top_k, logits.size(-1))
        logits = logits.masked_fill(logits < torch.topk(logits, k)[0][..., -1, None], float("-inf"))
    if top_p > 0:
        assert top_p <= 1.0
        if 0.0 < top_p < 1.0:
            sl, si = torch.sort(logits, descending=False)
            rm = sl.softmax(dim=-1).cumsum(dim=-1) <= (1 - top_p)
            logits = logits.masked_fill(rm.scatter(-1, si, rm), float("-inf"))
    return F.log_softmax(logits, dim=-1)


def atsp_policy(w: W, td0: dict, sidx: Tensor, num_starts: int, decode: str = "greedy",
                actions: Optional[Tensor] = None, trace: Optional[dict] = None, normalize: bool = True) -> dict:
    """RRNetPolicy.forward (policy.py:138-255) + pre/post decoder hooks (decoding.py:157-217)
    for ATSP.  `td0` is a *reset* state.  decode in {'greedy','evaluate'}; multistart iff
    num_starts > 1 (get_decoding_strategy: 'multistart_greedy').  With `trace` given, per-step
    logits / mask / logprobs are appended (golden traces)."""
    row, col = atsp_encoder(w, td0, sidx)
    if trace is not None:
        trace["row_emb"], trace["col_emb"] = row, col
    B, N = td0["action_mask"].shape
    S = num_starts if num_starts > 1 else 0
    acts, lps = [], []
    if S >= 1:
        a0 = torch.arange(S).repeat_interleave(B) % N  # select_start_nodes for atsp
        td = batchify_state({k: v for k, v in td0.items() if k not in ("locs",)}, S)
        td["action"] = a0
        td = atsp_step(td)
        lps.append(torch.zeros_like(a0, dtype=torch.float32))
        acts.append(a0)
    else:
        td = dict(td0)
    cache = precompute_cache(w, row, col)
    cache["_D"] = td0["distance_matrix"]
    k = 0
    while not td["done"].all():
        logits, mask = atsp_decoder_step(w, td, cache, S)
        logp = process_logits(logits, mask)
        if decode == "greedy":
            sel = logp.argmax(dim=-1)
        elif decode == "evaluate":
            sel = actions[:, k]   # policy.py:218: actions[..., step], step counts decode-loop iterations
        else:
            raise ValueError(decode)
        if trace is not None:
            trace.setdefault("logits", []).append(logits)
            trace.setdefault("mask", []).append(mask)
            trace.setdefault("logp", []).append(logp)
        lps.append(gather_by_index(logp, sel, dim=1))
        acts.append(sel)
        td["action"] = sel
        td = atsp_step(td)
        k += 1
    logprobs = torch.stack(lps, 1)
    actions_out = torch.stack(acts, 1)
    tdr = dict(td)
    out = {}
    if normalize:
        real, nd = atsp_reward(tdr, actions_out, True)
        out["reward"], out["normalized_reward"] = real, nd
    else:
        out["reward"] = atsp_reward(tdr, actions_out, False)
    assert (logprobs > -1000).all()
    out["log_likelihood"] = logprobs.sum(1)
    out["actions"] = actions_out
    out["logprobs"] = logprobs
    return out


def atsp_beam_search(w: W, td0: dict, sidx: Tensor, beam_width: int, select_best: bool = True) -> dict:
    """RRNetPolicy.forward with decode_type='beam_search' (decoding.py:402-554: BeamSearch with select_best=True) for ATSP."""
    row, col = atsp_encoder(w, td0, sidx)
    B, N = td0["action_mask"].shape
    W_ = beam_width
    a0 = torch.arange(W_).repeat_interleave(B) % N
    td = batchify_state({k: v for k, v in td0.items() if k not in ("locs",)}, W_)
    td["action"] = a0
    td = atsp_step(td)
    lp0 = torch.zeros_like(td["action_mask"], dtype=torch.float32)
    logps, acts, path = [lp0], [a0], [torch.zeros(B * W_, dtype=torch.int32)]
    parent_lp = lp0.gather(1, a0[..., None])
    cache = precompute_cache(w, row, col)
    cache["_D"] = td0["distance_matrix"]
    seq = torch.arange(B).repeat(W_)
    while not td["done"].all():
        logits, mask = atsp_decoder_step(w, td, cache, W_)
        logp = process_logits(logits, mask)
        stacked = torch.cat((logp + parent_lp).split(B), dim=1)
        top_lp, top_ix = torch.topk(stacked, W_, dim=1)
        parent_lp = torch.hstack(torch.unbind(top_lp, 1)).unsqueeze(1)
        top_ix = torch.hstack(torch.unbind(top_ix, 1))
        sel, par = top_ix % N, (top_ix // N).int()
        idx = seq + par * B
        path.append(par)
        td = {k: v[idx] for k, v in td.items()}
        assert not (~mask[idx]).gather(1, sel.unsqueeze(-1)).any()
        logps.append(logp[idx]); acts.append(sel)
        td["action"] = sel
        td = atsp_step(td)
    actions, logprobs = torch.stack(acts, 1), torch.stack(logps, 1)
    cur = path[-1]
    seqs, lps = [actions[:, -1]], [logprobs[:, -1]]
    for k in reversed(range(len(path) - 1)):
        idx = seq + cur * B
        seqs.append(actions[idx, k]); lps.append(logprobs[idx, k])
        cur = path[k][idx]
    actions, logprobs = torch.stack(list(reversed(seqs)), 1), torch.stack(list(reversed(lps)), 1)
    tdb = dict(td)
    if select_best:          # decoding.py:499-505 (on the real reward; the reference itself trips over the tuple here)
        real, nd = atsp_reward(tdb, actions, True)
        _, best = torch.cat(real.unsqueeze(1).split(B), 1).max(1)
        flat = torch.arange(B) + best * B
        actions, logprobs = actions[flat], logprobs[flat]
        tdb = {k: v[flat] for k, v in td.items()}
    real, nd = atsp_reward(tdb, actions, True)
    ll = logprobs.gather(-1, actions.unsqueeze(-1)).squeeze(-1).sum(1)
    return {"reward": real, "normalized_reward": nd, "log_likelihood": ll, "actions": actions}


# ----------------------------------------------------------------------------------------------
# Deterministic weights (inputs shared by oracle, reference and the HIP path)
# ----------------------------------------------------------------------------------------------
def make_weights(template: Dict[str, tuple], seed: int) -> W:
    """Reproducible stand-in for `nn.Linear` default init (U(-1/sqrt(fan_in), 1/sqrt(fan_in))),
    norm weights 1 + small noise, scalars (alpha/beta/temperature) perturbed around their
    defaults — drawn from numpy PCG64 keyed by (seed, parameter name) so the vectors do not
    depend on module construction order or torch's RNG stream."""
    import zlib

    import numpy as np
    out = {}
    for name, shape in template.items():
        rng = np.random.default_rng([seed, zlib.crc32(name.encode())])
        if name.endswith("normalizer.weight"):
            a = 1.0 + 0.1 * rng.standard_normal(shape)
        elif name.endswith("normalizer.bias") or name.endswith("normalizer.running_mean"):
            a = 0.1 * rng.standard_normal(shape)
        elif name.endswith("normalizer.running_var"):
            a = 1.0 + 0.3 * np.abs(rng.standard_normal(shape))
        elif name.endswith("num_batches_tracked"):
            a = np.asarray(100.0)
        elif name.endswith(".alpha") or name.endswith(".beta") or name.endswith("distance_weight") or name.endswith("duration_weight"):
            a = 1.0 + 0.1 * rng.standard_normal(shape)
        elif name.endswith("gate_temperature"):
            a = np.full(shape, 5.0) + 0.1 * rng.standard_normal(shape)
        elif name.endswith("W_placeholder"):
            a = rng.uniform(-1, 1, shape)
        else:
            fan_in = shape[-1] if len(shape) > 1 else None
            if fan_in is None:  # bias: fan_in of the sibling weight
                wname = name[: -len("bias")] + "weight"
                fan_in = template[wname][-1] if wname in template else shape[0]
            bound = 1.0 / math.sqrt(fan_in)
            a = rng.uniform(-bound, bound, shape)
        out[name] = torch.from_numpy(np.asarray(a, dtype=np.float32).reshape(shape)).clone()
    return out


def atsp_weight_template(embed_dim: int = 128, num_layers: int = 6, ff: int = 512, sample_size: int = 25) -> Dict[str, tuple]:
    """state_dict names/shapes of RRNetPolicy(env_name='atsp', ...) (probed from the reference;
    `oracle/gen_golden.py` asserts equality with the real module's state_dict)."""
    E = embed_dim
    t: Dict[str, tuple] = {}
    p = "encoder.init_embedding"
    t[p + ".init_embed.weight"] = (E, 2); t[p + ".init_embed.bias"] = (E,)
    for rc in ("row", "col"):
        t[f"{p}.{rc}_embed.weight"] = (E, sample_size); t[f"{p}.{rc}_embed.bias"] = (E,)
    for rc in ("row", "col"):
        q = f"{p}.gating_network_{rc}.gating_fc"
        t[q + ".0.weight"] = (2 * E, 2 * E); t[q + ".0.bias"] = (2 * E,)
        t[q + ".2.weight"] = (1, 2 * E); t[q + ".2.bias"] = (1,)
    for l in range(num_layers):
        for rc in ("row", "col"):
            b = f"encoder.net.layers.{l}.{rc}_encoding_block"
            t[b + ".alpha"] = (1,)
            for nm in ("to_q", "to_k", "to_v", "project"):
                t[f"{b}.attn_free.{nm}.weight"] = (E, E); t[f"{b}.attn_free.{nm}.bias"] = (E,)
            t[b + ".multi_head_combine.weight"] = (E, E); t[b + ".multi_head_combine.bias"] = (E,)
            f = b + ".angle_distance_fusion"
            for nm in ("dist_emb", "angle_emb"):
                t[f"{f}.{nm}.0.weight"] = (E, 1); t[f"{f}.{nm}.0.bias"] = (E,)
                t[f"{f}.{nm}.2.weight"] = (E, E); t[f"{f}.{nm}.2.bias"] = (E,)
            t[f + ".gate.0.weight"] = (1, 2 * E); t[f + ".gate.0.bias"] = (1,)
            t[f + ".out_lin.weight"] = (1, E); t[f + ".out_lin.bias"] = (1,)
            o = b + ".feed_forward.ops"
            t[o + ".norm1.normalizer.weight"] = (E,); t[o + ".norm1.normalizer.bias"] = (E,)
            t[o + ".ffn.W1.weight"] = (ff, E); t[o + ".ffn.W1.bias"] = (ff,)
            t[o + ".ffn.W2.weight"] = (E, ff); t[o + ".ffn.W2.bias"] = (E,)
            t[o + ".norm2.normalizer.weight"] = (E,); t[o + ".norm2.normalizer.bias"] = (E,)
            for nm in ("norm1", "norm2", "norm3"):
                t[f"{b}.{nm}.normalizer.weight"] = (E,); t[f"{b}.{nm}.normalizer.bias"] = (E,)
    t["decoder.alpha"] = (1,)
    t["decoder.context_embedding.W_placeholder"] = (2 * E,)
    t["decoder.context_embedding.project_context.weight"] = (E, 2 * E)
    t["decoder.pointer.project_out.weight"] = (E, E)
    t["decoder.pointer.ffn.lins.0.weight"] = (4 * E, E); t["decoder.pointer.ffn.lins.0.bias"] = (4 * E,)
    t["decoder.pointer.ffn.lins.1.weight"] = (E, 4 * E); t["decoder.pointer.ffn.lins.1.bias"] = (E,)
    t["decoder.project_node_embeddings.weight"] = (3 * E, E)
    t["decoder.project_fixed_context.weight"] = (E, E)
    return t


# ==============================================================================================
# RCVRP  (rrnco/envs/rcvrp/env.py, rrnco/models/env_embeddings/rcvrp.py, rl4co VRPContext)
# ==============================================================================================
def rcvrp_synthetic(batch: int, n: int, seed: int, capacity: float = 50.0) -> dict:
    """Synthetic RCVRP instance: uniform depot/customers (rcvrp/generator_lazy.py:244-257),
    an asymmetric matrix D = cdist * (1 + 0.2 U) with zero diagonal supplied explicitly (the env's Euclidean
    fallback has the wrong shape), integer demands 1..9 / capacity
    (scripts/generate_data.py:206-207)."""
    g = torch.Generator().manual_seed(seed)
    depot = torch.rand(batch, 2, generator=g)
    locs = torch.rand(batch, n, 2, generator=g)
    pts = torch.cat([depot[:, None], locs], 1)
    D = torch.cdist(pts, pts) * (1 + 0.2 * torch.rand(batch, n + 1, n + 1, generator=g))
    D[:, torch.arange(n + 1), torch.arange(n + 1)] = 0
    demand = torch.randint(1, 10, (batch, n), generator=g).float() / capacity
    return {"locs": locs, "depot": depot, "distance_matrix": D, "demand": demand}


def rcvrp_action_mask(td: dict) -> Tensor:
    """env.py:183-195."""
    exceeds = td["demand"] + td["used_capacity"] > td["vehicle_capacity"]
    mask_loc = td["visited"][..., 1:].to(exceeds.dtype) | exceeds
    mask_depot = (td["current_node"] == 0) & ((mask_loc == 0).int().sum(-1) > 0)[:, None]
    return ~torch.cat((mask_depot, mask_loc), -1)


def rcvrp_reset(td: dict, vehicle_capacity: float = 1.0, normalize: bool = True) -> dict:
    """env.py:124-181."""
    distance = td["distance_matrix"]
    B = distance.shape[0]
    out = {}
    if normalize:
        mn = distance.amin(dim=(-2, -1), keepdim=True)
        mx = distance.amax(dim=(-2, -1), keepdim=True)
        distance = ((distance - mn) / (mx - mn + 1e-6)).to(torch.float32)
        out["min_distance"], out["max_distance"] = mn.squeeze(-1).squeeze(-1), mx.squeeze(-1).squeeze(-1)
    depot = td["depot"].unsqueeze(1) if td["depot"].ndim == 2 else td["depot"]
    out.update(
        locs=torch.cat((depot, td["locs"]), dim=-2), distance_matrix=distance, demand=td["demand"],
        current_node=torch.zeros(B, 1, dtype=torch.long), used_capacity=torch.zeros(B, 1),
        vehicle_capacity=torch.full((B, 1), vehicle_capacity),
        visited=torch.zeros(B, td["locs"].shape[-2] + 1, dtype=torch.uint8),
        done=torch.zeros(B, 1, dtype=torch.bool))
    out["action_mask"] = rcvrp_action_mask(out)
    return out


def rcvrp_step(td: dict) -> dict:
    """env.py:90-122."""
    cur = td["action"][:, None]
    n_loc = td["demand"].size(-1)
    sel = gather_by_index(td["demand"], torch.clamp(cur - 1, 0, n_loc - 1), squeeze=False)
    used = (td["used_capacity"] + sel) * (cur != 0).float()
    visited = td["visited"].scatter(-1, cur, 1)
    done = visited.sum(-1) == visited.size(-1)
    td.update(current_node=cur, used_capacity=used, visited=visited, reward=torch.zeros_like(done), done=done)
    td["action_mask"] = rcvrp_action_mask(td)
    return td


def vrp_reward(td: dict, actions: Tensor, normalize: bool = True):
    """rcvrp/env.py:197-219 (depot-prefixed, rolled)."""
    D = td["distance_matrix"]
    go_from = torch.cat((torch.zeros_like(actions[:, :1]), actions), dim=1)
    go_to = torch.roll(go_from, -1, dims=1)
    dist = gather_by_index(gather_by_index(D, go_from, dim=1, squeeze=False), go_to, dim=2, squeeze=False).squeeze(-1)
    if normalize:
        nd = -dist.sum(-1)
        return nd * (td["max_distance"] - td["min_distance"] + 1e-6) + td["min_distance"], nd
    return -dist.sum(-1)


def rcvrp_check(td: dict, actions: Tensor) -> bool:
    """env.py:221-249."""
    B, n = td["demand"].size()
    sp = actions.sort(1)[0]
    ok = bool((torch.arange(1, n + 1).view(1, -1).expand(B, n) == sp[:, -n:]).all() and (sp[:, :-n] == 0).all())
    d = torch.cat((-td["vehicle_capacity"], td["demand"]), 1).gather(1, actions)
    used = torch.zeros_like(td["demand"][:, 0])
    for i in range(actions.size(1)):
        used = used + d[:, i]
        used[used < 0] = 0
        ok = ok and bool((used <= td["vehicle_capacity"][:, 0] + 1e-5).all())
    return ok


def rcvrp_init_embedding(w: W, locs: Tensor, demand: Tensor, distance: Tensor, sidx: Tensor, extra_feats: Optional[Tensor] = None):
    """RVRPInitEmbedding._embed_with_distance rcvrp.py:88-102 (+ CoordinateExpert :105-124, DistanceExpert :127-150);
    rcvrptw.py differs only in demand_init taking (demand, tw0, tw1, service) (`extra_feats`)."""
    p = "encoder.init_embedding"
    locs = locs.float()
    depot, cities = locs[:, :1, :], locs[:, 1:, :]
    c = cities - depot
    ang = torch.atan2(c[..., 1:], c[..., :1])
    # depot is a strided slice [B,1,2]: F.linear on a NON-contiguous input takes a different CPU path for plain tensors
    # than for the reference module's Parameters (requires_grad) — 1 ulp apart; on a contiguous input both take the same one
    node = torch.cat([lin(w, p + ".coord_expert.init_embed_depot", depot.contiguous()),
                      lin(w, p + ".coord_expert.init_embed", torch.cat([cities, ang], dim=-1))], dim=-2)
    rowd = distance.gather(2, sidx)
    cold = distance.transpose(1, 2).gather(2, sidx)
    row = lin(w, p + ".distance_expert.row_embed", rowd.sort(dim=-1).values)
    col = lin(w, p + ".distance_expert.col_embed", cold.sort(dim=-1).values)
    crow = contextual_gating(w, p + ".gating_network_row", node, row)
    ccol = contextual_gating(w, p + ".gating_network_col", node, col)
    dem = torch.cat([torch.zeros_like(demand[:, :1]), demand], dim=1)[..., None]
    feats = dem if extra_feats is None else torch.cat([dem, extra_feats], -1)
    de = lin(w, p + ".demand_init", feats)
    return (lin(w, p + ".combine_row_embed", torch.cat([crow, de], -1)),
            lin(w, p + ".combine_col_embed", torch.cat([ccol, de], -1)))


def rcvrp_decoder_step(w: W, td_flat: dict, cache: dict, S: int):
    """RRNetDecoder.forward for rcvrp (decoder.py:151-206) with rl4co VRPContext (E+1 -> E)."""
    keys = ("current_node", "used_capacity", "vehicle_capacity", "action_mask")
    td = {k: (unbatchify(td_flat[k], S) if S > 1 else td_flat[k]) for k in keys}
    emb = cache["node_embeddings"]
    cur = gather_by_index(emb, td["current_node"])
    q = F.linear(torch.cat([cur, td["vehicle_capacity"] - td["used_capacity"]], -1),
                 w["decoder.context_embedding.project_context.weight"])
    q = q.unsqueeze(1) if q.ndim == 2 else q
    mask = td["action_mask"]
    logits = pointer(w, q, cache["glimpse_key"], cache["glimpse_val"], cache["logit_key"], mask)
    D = cache["_D"]
    if S > 1:
        bias = w["decoder.alpha"] * gather_by_index(D.unsqueeze(1).expand(-1, S, -1, -1), td["current_node"], dim=-2)
    else:
        bias = w["decoder.alpha"] * gather_by_index(D, td["current_node"], dim=-2)
    _ft = torch.float64 if logits.dtype == torch.float64 else torch.float32      # (decoder.py:195-196 casts to fp32; a float64 run of the oracle stays float64)
    logits = torch.log(torch.exp(logits.to(_ft) - bias.to(_ft)) + 1e-6)
    if S > 1:
        logits = logits.permute(1, 0, 2).reshape(-1, logits.shape[-1])
        mask = mask.permute(1, 0, 2).reshape(-1, mask.shape[-1])
    return logits, mask


def rcvrp_policy(w: W, td0: dict, sidx: Tensor, num_starts: int, decode: str = "greedy",
                 actions: Optional[Tensor] = None, trace: Optional[dict] = None) -> dict:
    """RRNetPolicy.forward for RCVRP; td0 = rcvrp_reset(...).  S = N+1 starts as test.py:132 uses."""
    row, col = rcvrp_init_embedding(w, td0["locs"], td0["demand"], td0["distance_matrix"], sidx)
    row, col = encoder_net(w, row, col, td0["distance_matrix"], td0["locs"].float(), None, num_layers_of(w))
    if trace is not None:
        trace["row_emb"], trace["col_emb"] = row, col
    B, N1 = td0["action_mask"].shape
    n_loc = N1 - 1
    S = num_starts if num_starts > 1 else 0
    acts, lps = [], []
    static = ("locs", "distance_matrix", "min_distance", "max_distance")
    if S >= 1:
        a0 = torch.arange(S).repeat_interleave(B) % n_loc + 1
        td = batchify_state({k: v for k, v in td0.items() if k not in static}, S)
        td["action"] = a0
        td = rcvrp_step(td)
        lps.append(torch.zeros_like(a0, dtype=torch.float32)); acts.append(a0)
    else:
        td = {k: v for k, v in td0.items() if k not in static}
    cache = precompute_cache(w, row, col)
    cache["_D"] = td0["distance_matrix"]
    k = 0
    while not td["done"].all():
        logits, mask = rcvrp_decoder_step(w, td, cache, S)
        logp = process_logits(logits, mask)
        sel = logp.argmax(dim=-1) if decode == "greedy" else actions[:, k]
        if trace is not None:
            trace.setdefault("logits", []).append(logits); trace.setdefault("mask", []).append(mask)
            trace.setdefault("logp", []).append(logp)
        lps.append(gather_by_index(logp, sel, dim=1)); acts.append(sel)
        td["action"] = sel
        td = rcvrp_step(td)
        k += 1
    logprobs, actions_out = torch.stack(lps, 1), torch.stack(acts, 1)
    R = actions_out.shape[0]
    rtd = {"distance_matrix": td0["distance_matrix"][torch.arange(R) % B],
           "min_distance": td0["min_distance"][torch.arange(R) % B], "max_distance": td0["max_distance"][torch.arange(R) % B]}
    real, nd = vrp_reward(rtd, actions_out, True)
    return {"reward": real, "normalized_reward": nd, "log_likelihood": logprobs.sum(1), "actions": actions_out,
            "logprobs": logprobs}


def rcvrp_weight_template(embed_dim: int = 128, num_layers: int = 6, ff: int = 512, sample_size: int = 25,
                          demand_feats: int = 1, ctx_state: int = 1) -> Dict[str, tuple]:
    E = embed_dim
    t = {k: v for k, v in atsp_weight_template(E, num_layers, ff, sample_size).items()
         if not k.startswith("encoder.init_embedding") and "context_embedding" not in k}
    p = "encoder.init_embedding"
    t[p + ".coord_expert.init_embed_depot.weight"] = (E, 2); t[p + ".coord_expert.init_embed_depot.bias"] = (E,)
    t[p + ".coord_expert.init_embed.weight"] = (E, 3); t[p + ".coord_expert.init_embed.bias"] = (E,)
    for rc in ("row", "col"):
        q = f"{p}.gating_network_{rc}.gating_fc"
        t[q + ".0.weight"] = (2 * E, 2 * E); t[q + ".0.bias"] = (2 * E,)
        t[q + ".2.weight"] = (1, 2 * E); t[q + ".2.bias"] = (1,)
        t[f"{p}.combine_{rc}_embed.weight"] = (E, 2 * E); t[f"{p}.combine_{rc}_embed.bias"] = (E,)
        t[f"{p}.distance_expert.{rc}_embed.weight"] = (E, sample_size); t[f"{p}.distance_expert.{rc}_embed.bias"] = (E,)
        t[f"{p}.distance_expert.{rc}_combine_embed.weight"] = (E, 2 * E); t[f"{p}.distance_expert.{rc}_combine_embed.bias"] = (E,)
    t[p + ".demand_init.weight"] = (E, demand_feats); t[p + ".demand_init.bias"] = (E,)
    t["decoder.context_embedding.project_context.weight"] = (E, E + ctx_state)
    return t


# ==============================================================================================
# RCVRPTW = RMTVRPEnv with the vrptw preset (rrnco/envs/rmtvrp/env.py, configs/env/rcvrptw.yaml)
# ==============================================================================================
def vrptw_capacity(n: int) -> float:
    """rmtvrp/generator.py:20-33."""
    return 30.0 + (n // 5 if n > 20 else 0)


def rcvrptw_synthetic(batch: int, n: int, seed: int, max_time: float = 4.6) -> dict:
    """Synthetic VRPTW instance: LazyRMTVRPGenerator(variant_preset='vrptw') synthetic branch
    (rmtvrp/generator_lazy.py:302-348): uniform locs incl. depot, integer linehaul demands 1..9 / capacity
    (generator.py:445-469 with backhaul_ratio 0, scale_demand), time windows / service times of
    generator.py:495-513 (speed 1), plus explicit asymmetric distance and min-max normalised duration
    matrices (generator.py:246-262 normalises durations) so that D != Dur != D^T."""
    g = torch.Generator().manual_seed(seed)
    locs = torch.rand(batch, n + 1, 2, generator=g)
    cd = torch.cdist(locs, locs)
    D = cd * (1 + 0.2 * torch.rand(batch, n + 1, n + 1, generator=g))
    T = cd * (1 + 0.3 * torch.rand(batch, n + 1, n + 1, generator=g))
    ar = torch.arange(n + 1)
    D[:, ar, ar] = 0; T[:, ar, ar] = 0
    tmn, tmx = T.amin(dim=(1, 2), keepdim=True), T.amax(dim=(1, 2), keepdim=True)
    T = (T - tmn) / (tmx - tmn)
    demand = (torch.rand(batch, n, generator=g) * 9).int().add(1).float() / vrptw_capacity(n)
    a, b, c = 0.15, 0.18, 0.2
    service = a + (b - a) * torch.rand(batch, n, generator=g)
    tw_len = b + (c - b) * torch.rand(batch, n, generator=g)
    d0 = (locs[:, 0:1] - locs[:, 1:]).norm(p=2, dim=-1)
    h_max = (max_time - service - tw_len) / d0 - 1
    tw_start = (1 + (h_max - 1) * torch.rand(batch, n, generator=g)) * d0
    tw_end = tw_start + tw_len
    tw = torch.stack((torch.cat((torch.zeros(batch, 1), tw_start), -1),
                      torch.cat((torch.full((batch, 1), max_time), tw_end), -1)), dim=-1)
    service = torch.cat((torch.zeros(batch, 1), service), dim=-1)
    return {"locs": locs, "distance_matrix": D, "duration_matrix": T, "demand_linehaul": demand,
            "time_windows": tw, "service_time": service}


def rmtvrp_variant_synthetic(batch: int, n: int, seed: int) -> dict:
    """A mixed-variant MTVRP batch on top of rcvrptw_synthetic: about a quarter of the customers are backhauls, backhaul
    class 1 (classical) or 2 (mixed) per instance, open routes on some instances, a finite distance limit on some (>= 2.2 on
    the normalised matrix, so every customer can be served on a route of its own)."""
    inst = rcvrptw_synthetic(batch, n, seed)
    g = torch.Generator().manual_seed(seed + 7919)
    back = torch.rand(batch, n, generator=g) < 0.25
    dem = inst["demand_linehaul"]
    inst["demand_backhaul"] = torch.where(back, dem, torch.zeros_like(dem))
    inst["demand_linehaul"] = torch.where(back, torch.zeros_like(dem), dem)
    inst["backhaul_class"] = torch.randint(1, 3, (batch, 1), generator=g).to(torch.int32)
    inst["open_route"] = torch.rand(batch, 1, generator=g) < 0.5
    lim = 2.2 + 1.3 * torch.rand(batch, 1, generator=g)
    inst["distance_limit"] = torch.where(torch.rand(batch, 1, generator=g) < 0.5, lim, torch.full_like(lim, float("inf")))
    return inst


def rmtvrp_action_mask(td: dict) -> Tensor:
    """rmtvrp/env.py:343-428 (all 16-variant terms computed; O/L/B/MB are inert under the vrptw preset)."""
    cur = td["current_node"]
    B = cur.shape[0]
    bi = torch.arange(B)
    D, T = td["distance_matrix"], td["duration_matrix"]
    dist_ij, dist_j0 = D[bi, cur, :], D[:, :, 0]
    dur_ij, dur_j0 = T[bi, cur, :], T[:, :, 0]
    early, late = td["time_windows"][..., 0], td["time_windows"][..., 1]
    arrival = td["current_time"] + dur_ij
    can_reach_customer = arrival < late
    can_reach_depot = (torch.max(arrival, early) + td["service_time"] + dur_j0) * ~td["open_route"] < late[..., 0:1]
    exceeds_dist = td["current_route_length"] + dist_ij + (dist_j0 * ~td["open_route"]) > td["distance_limit"]
    ex_l = td["demand_linehaul"] + td["used_capacity_linehaul"] > td["vehicle_capacity"]
    ex_b = td["demand_backhaul"] + td["used_capacity_backhaul"] > td["vehicle_capacity"]
    line_missing = ((td["demand_linehaul"] * ~td["visited"]).sum(-1) > 0)[..., None]
    carrying_b = gather_by_index(td["demand_backhaul"], cur, dim=1, squeeze=False) > 0
    ok1 = (line_missing & ~ex_l & ~carrying_b & (td["demand_linehaul"] > 0)) | (~ex_b & (td["demand_backhaul"] > 0))
    cannot_l = td["demand_linehaul"] > td["vehicle_capacity"] - td["used_capacity_backhaul"]
    ok2 = ~ex_l & ~ex_b & ~cannot_l
    ok = ((td["backhaul_class"] == 1) & ok1) | ((td["backhaul_class"] == 2) & ok2)
    can = can_reach_customer & can_reach_depot & ok & ~exceeds_dist & ~td["visited"]
    can[:, 0] = ~((cur == 0) & (can[:, 1:].sum(-1) > 0))
    return can


def rmtvrp_reset(td: dict, normalize: bool = True) -> dict:
    """rmtvrp/env.py:217-341."""
    B = td["locs"].shape[0]
    dl = torch.cat([torch.zeros_like(td["demand_linehaul"][..., :1]), td["demand_linehaul"]], dim=1)
    db = torch.cat([torch.zeros_like(td["demand_linehaul"][..., :1]),
                    td.get("demand_backhaul", torch.zeros_like(td["demand_linehaul"]))], dim=1)
    D = td["distance_matrix"]
    out = {}
    if normalize:
        mn, mx = D.amin(dim=(-2, -1), keepdim=True), D.amax(dim=(-2, -1), keepdim=True)
        D = ((D - mn) / (mx - mn + 1e-6)).to(torch.float32)
        out["min_distance"], out["max_distance"] = mn.squeeze(-1).squeeze(-1), mx.squeeze(-1).squeeze(-1)
    speed = torch.ones_like(dl[..., :1])
    out.update(
        locs=td["locs"], distance_matrix=D, duration_matrix=td.get("duration_matrix", D / speed[:, None]),
        demand_backhaul=db, demand_linehaul=dl,
        backhaul_class=td.get("backhaul_class", torch.full((B, 1), 1, dtype=torch.int32)),                # env.py:236-243
        distance_limit=td.get("distance_limit", torch.full_like(dl[..., :1], float("inf"))),             # :254-257
        service_time=td["service_time"],
        open_route=td.get("open_route", torch.zeros_like(dl[..., :1], dtype=torch.bool)),                # :249-252
        time_windows=td["time_windows"], speed=speed,
        vehicle_capacity=torch.ones_like(dl[..., :1]), capacity_original=torch.ones_like(dl[..., :1]),
        current_node=torch.zeros(B, dtype=torch.long), current_route_length=torch.zeros(B, 1),
        current_time=torch.zeros(B, 1), used_capacity_backhaul=torch.zeros(B, 1), used_capacity_linehaul=torch.zeros(B, 1),
        visited=torch.zeros(B, td["locs"].shape[-2], dtype=torch.bool), done=torch.zeros(B, 1, dtype=torch.bool))
    out["action_mask"] = rmtvrp_action_mask(out)
    return out


def rmtvrp_step(td: dict) -> dict:
    """rmtvrp/env.py:155-215."""
    prev, cur = td["current_node"], td["action"]
    bi = torch.arange(cur.shape[0])
    dist = td["distance_matrix"][bi, prev, cur]
    dur = td["duration_matrix"][bi, prev, cur]
    service = gather_by_index(td["service_time"], cur, dim=1, squeeze=False)
    start = gather_by_index(td["time_windows"], cur, dim=1, squeeze=False)[..., 0]
    nz = cur[:, None] != 0
    ctime = nz * (torch.max(td["current_time"] + dur[:, None], start) + service)
    clen = nz * (td["current_route_length"] + dist[:, None])
    ul = nz * (td["used_capacity_linehaul"] + gather_by_index(td["demand_linehaul"], cur, dim=1, squeeze=False))
    ub = nz * (td["used_capacity_backhaul"] + gather_by_index(td["demand_backhaul"], cur, dim=1, squeeze=False))
    visited = td["visited"].scatter(-1, cur[..., None], True)
    done = visited.sum(-1) == visited.size(-1)
    td.update(current_node=cur, current_route_length=clen, current_time=ctime, done=done,
              reward=torch.zeros_like(done).float(), used_capacity_linehaul=ul, used_capacity_backhaul=ub, visited=visited)
    td["action_mask"] = rmtvrp_action_mask(td)
    return td


def mtvrp_context(w: W, emb: Tensor, td: dict) -> Tensor:
    """MTVRPContextEmbedding env_embeddings/context.py:34-70 (Linear(E+4, E, bias=False))."""
    cur = gather_by_index(emb, td["current_node"])
    used = torch.where(td["used_capacity_backhaul"] == 0, td["used_capacity_linehaul"], td["used_capacity_backhaul"])
    rem = torch.nan_to_num(td["distance_limit"] - td["current_route_length"], posinf=10)
    feats = torch.cat((td["vehicle_capacity"] - used, td["current_time"], td["open_route"].float(), rem), -1)
    return F.linear(torch.cat([cur, feats], -1), w["decoder.context_embedding.project_context.weight"])


RMTVRP_DYN = ("current_node", "used_capacity_backhaul", "used_capacity_linehaul", "vehicle_capacity", "current_time",
              "open_route", "distance_limit", "current_route_length", "action_mask")


def rcvrptw_decoder_step(w: W, td_flat: dict, cache: dict, S: int):
    """RRNetDecoder.forward for rcvrptw (decoder.py:151-206): bias = alpha*D[cur,:] + beta*Dur[cur,:]."""
    td = {k: (unbatchify(td_flat[k], S) if S > 1 else td_flat[k]) for k in RMTVRP_DYN}
    q = mtvrp_context(w, cache["node_embeddings"], td)
    q = q.unsqueeze(1) if q.ndim == 2 else q
    mask = td["action_mask"]
    logits = pointer(w, q, cache["glimpse_key"], cache["glimpse_val"], cache["logit_key"], mask)
    D, T = cache["_D"], cache["_T"]
    if S > 1:
        dist = gather_by_index(D.unsqueeze(1).expand(-1, S, -1, -1), td["current_node"], dim=-2)
        dur = gather_by_index(T.unsqueeze(1).expand(-1, S, -1, -1), td["current_node"], dim=-2)
    else:
        dist = gather_by_index(D, td["current_node"], dim=-2)
        dur = gather_by_index(T, td["current_node"], dim=-2)
    bias = w["decoder.alpha"] * dist + w["decoder.beta"] * dur
    _ft = torch.float64 if logits.dtype == torch.float64 else torch.float32      # (decoder.py:195-196 casts to fp32; a float64 run of the oracle stays float64)
    logits = torch.log(torch.exp(logits.to(_ft) - bias.to(_ft)) + 1e-6)
    if S > 1:
        logits = logits.permute(1, 0, 2).reshape(-1, logits.shape[-1])
        mask = mask.permute(1, 0, 2).reshape(-1, mask.shape[-1])
    return logits, mask


def rcvrptw_policy(w: W, td0: dict, sidx: Tensor, num_starts: int, decode: str = "greedy",
                   actions: Optional[Tensor] = None, trace: Optional[dict] = None) -> dict:
    """RRNetPolicy.forward for RCVRPTW; td0 = rmtvrp_reset(...); S = N starts (test.py:129-131)."""
    feats = torch.cat([td0["time_windows"], td0["service_time"][..., None]], -1)
    row, col = rcvrp_init_embedding(_tw_init_names(w), td0["locs"], td0["demand_linehaul"][:, 1:], td0["distance_matrix"], sidx, feats)
    row, col = encoder_net(w, row, col, td0["distance_matrix"], td0["locs"].float(), td0["duration_matrix"].float(), num_layers_of(w))
    if trace is not None:
        trace["row_emb"], trace["col_emb"] = row, col
    B, N1 = td0["action_mask"].shape
    S = num_starts if num_starts > 1 else 0
    static = ("locs", "distance_matrix", "duration_matrix", "min_distance", "max_distance")
    acts, lps = [], []
    if S >= 1:
        a0 = torch.arange(S).repeat_interleave(B) % (N1 - 1) + 1      # AllSelectStartNodes selectstartnodes.py:42-50
        td = batchify_state({k: v for k, v in td0.items() if k not in static}, S)
        td["distance_matrix"], td["duration_matrix"] = batchify(td0["distance_matrix"], S), batchify(td0["duration_matrix"], S)
        td["action"] = a0
        td = rmtvrp_step(td)
        lps.append(torch.zeros_like(a0, dtype=torch.float32)); acts.append(a0)
    else:
        td = {k: v for k, v in td0.items() if k != "locs"}
    cache = precompute_cache(w, row, col)
    cache["_D"], cache["_T"] = td0["distance_matrix"], td0["duration_matrix"]
    k = 0
    while not td["done"].all():
        logits, mask = rcvrptw_decoder_step(w, td, cache, S)
        logp = process_logits(logits, mask)
        sel = logp.argmax(dim=-1) if decode == "greedy" else actions[:, k]
        if trace is not None:
            trace.setdefault("logits", []).append(logits); trace.setdefault("mask", []).append(mask)
            trace.setdefault("logp", []).append(logp)
        lps.append(gather_by_index(logp, sel, dim=1)); acts.append(sel)
        td["action"] = sel
        td = rmtvrp_step(td)
        k += 1
    logprobs, actions_out = torch.stack(lps, 1), torch.stack(acts, 1)
    R = actions_out.shape[0]
    ix = torch.arange(R) % B
    Dr = td0["distance_matrix"].clone()
    Dr[:, :, 0] = Dr[:, :, 0] * ~td0["open_route"]     # rmtvrp/env.py:430-434: arcs into the depot are free on open routes
    rtd = {"distance_matrix": Dr[ix], "min_distance": td0["min_distance"][ix], "max_distance": td0["max_distance"][ix]}
    real, nd = vrp_reward(rtd, actions_out, True)      # rmtvrp/env.py:435-455
    return {"reward": real, "normalized_reward": nd, "log_likelihood": logprobs.sum(1), "actions": actions_out,
            "logprobs": logprobs}


def _tw_init_names(w: W) -> W:
    """rcvrptw.py names the attribute layer `init_embed` (Linear(4,E)) where rcvrp.py has `demand_init`."""
    out = dict(w)
    out["encoder.init_embedding.demand_init.weight"] = w["encoder.init_embedding.init_embed.weight"]
    out["encoder.init_embedding.demand_init.bias"] = w["encoder.init_embedding.init_embed.bias"]
    return out


def rcvrptw_weight_template(embed_dim: int = 128, num_layers: int = 6, ff: int = 512, sample_size: int = 25) -> Dict[str, tuple]:
    E = embed_dim
    t = {}
    for k, v in rcvrp_weight_template(E, num_layers, ff, sample_size, 4, 4).items():
        if ".angle_distance_fusion." in k:
            if ".gate.0." in k:
                continue
            k = k.replace(".angle_distance_fusion.", ".neural_adaptive_bias.")
        k = k.replace("encoder.init_embedding.demand_init.", "encoder.init_embedding.init_embed.")
        t[k] = v
    for l in range(num_layers):
        for rc in ("row", "col"):
            f = f"encoder.net.layers.{l}.{rc}_encoding_block.neural_adaptive_bias"
            t[f + ".dur_emb.0.weight"] = (E, 1); t[f + ".dur_emb.0.bias"] = (E,)
            t[f + ".dur_emb.2.weight"] = (E, E); t[f + ".dur_emb.2.bias"] = (E,)
            t[f + ".gate.0.weight"] = (E, 3 * E); t[f + ".gate.0.bias"] = (E,)
            t[f + ".gate.2.weight"] = (3, E); t[f + ".gate.2.bias"] = (3,)
            t[f + ".gate_temperature"] = ()
    t["decoder.beta"] = (1,)
    return t


# ----------------------------------------------------------------------------------------------
# MatNet baseline encoder (rrnco/baselines/MatNet/encoder.py).  Mixed-score attention = softmax attention
# whose per-head logit is a 2 -> 16 -> 1 MLP of (scaled dot product, distance-matrix entry).
# ----------------------------------------------------------------------------------------------
def matnet_weight_template(embed_dim: int = 256, heads: int = 16, layers: int = 5, ff: int = 512, env_name: str = "atsp",
                           mixer_hidden: int = 16) -> Dict[str, tuple]:
    """state_dict names / shapes of MatNetEncoder (encoder.py:175-215; configs/experiment/matnet.yaml: 256 / 16 / 5)."""
    E = embed_dim
    t = {}
    for l in range(layers):
        for side in ("row", "col"):
            p = f"layers.{l}.MHA.{side}_encoding_block"
            t[p + ".sdpa_fn.mix_W1"] = (heads, 2, mixer_hidden); t[p + ".sdpa_fn.mix_b1"] = (heads, mixer_hidden)
            t[p + ".sdpa_fn.mix_W2"] = (heads, mixer_hidden, 1); t[p + ".sdpa_fn.mix_b2"] = (heads, 1)
            t[p + ".Wq.weight"] = (E, E); t[p + ".Wkv.weight"] = (2 * E, E); t[p + ".out_proj.weight"] = (E, E)
        for f in ("F_a", "F_b"):
            p = f"layers.{l}.{f}.ops"
            t[p + ".norm1.normalizer.weight"] = (E,); t[p + ".norm1.normalizer.bias"] = (E,)
            t[p + ".ffn.W1.weight"] = (ff, E); t[p + ".ffn.W1.bias"] = (ff,)
            t[p + ".ffn.W2.weight"] = (E, ff); t[p + ".ffn.W2.bias"] = (E,)
            t[p + ".norm2.normalizer.weight"] = (E,); t[p + ".norm2.normalizer.bias"] = (E,)
    if env_name == "rcvrp":      # env_embeddings/rcvrp.py:22-33 with use_coords=False (configs/experiment/matnet.yaml:27-29)
        t["init_embedding.depot_client_emb.weight"] = (2, E)
        t["init_embedding.init_embed.weight"] = (E, 1); t["init_embedding.init_embed.bias"] = (E,)
        t["init_embedding.row_combine_embed.weight"] = (E, 2 * E); t["init_embedding.row_combine_embed.bias"] = (E,)
        t["init_embedding.col_combine_embed.weight"] = (E, 2 * E); t["init_embedding.col_combine_embed.bias"] = (E,)
    return t


def matnet_init_embedding(w: W, env_name: str, td: dict, rand_idx: Tensor, embed_dim: int):
    """env_embeddings/atsp.py:21-34 (zero rows, one-hot columns at a random permutation of feature slots) and
    env_embeddings/rcvrp.py:37-81 with use_coords=False.  `rand_idx` [B, N] is the reference's `rand.argsort(dim=1)`."""
    D = td["distance_matrix"]
    b, r, c = D.shape
    row = torch.zeros(b, r, embed_dim)
    col = torch.zeros(b, c, embed_dim)
    col[torch.arange(b)[:, None].expand(b, c), torch.arange(c)[None, :].expand(b, c), rand_idx] = 1.0
    if env_name == "atsp":
        return row, col, D
    p = "init_embedding"
    emb = w[p + ".depot_client_emb.weight"]                                          # rcvrp.py:70-76
    depot = emb[0:1].unsqueeze(1).expand(b, -1, -1)
    nodes = emb[1:2] + F.linear(td["demand"][..., None], w[p + ".init_embed.weight"], w[p + ".init_embed.bias"])
    out = torch.cat((depot, nodes), -2)                                              # :78
    row = F.linear(torch.cat([row, out], -1), w[p + ".row_combine_embed.weight"], w[p + ".row_combine_embed.bias"])
    col = F.linear(torch.cat([col, out], -1), w[p + ".col_combine_embed.weight"], w[p + ".col_combine_embed.bias"])
    return row, col, D


def matnet_cross_mha(w: W, p: str, q_in: Tensor, kv_in: Tensor, dmat: Tensor, heads: int) -> Tensor:
    """MatNetCrossMHA = rl4co MultiHeadCrossAttention [recalled: Wq, Wkv = (K | V) chunks, out_proj, no bias] around
    MixedScoresSDPA.forward (encoder.py:45-92)."""
    b, m, E = q_in.shape
    n = kv_in.shape[1]
    d = E // heads
    q = F.linear(q_in, w[p + ".Wq.weight"]).view(b, m, heads, d).permute(0, 2, 1, 3)
    kv = F.linear(kv_in, w[p + ".Wkv.weight"]).view(b, n, 2, heads, d).permute(2, 0, 3, 1, 4)
    k, v = kv[0], kv[1]
    s = torch.matmul(q, k.transpose(-2, -1)) / (d ** 0.5)                            # :52
    mix = torch.cat([s.unsqueeze(-1), dmat[:, None, :, :, None].expand(b, heads, m, n, 1)], -1)     # :54-60
    W1, b1 = w[p + ".sdpa_fn.mix_W1"], w[p + ".sdpa_fn.mix_b1"]
    W2, b2 = w[p + ".sdpa_fn.mix_W2"], w[p + ".sdpa_fn.mix_b2"]
    hid = F.relu(torch.matmul(mix.transpose(1, 2), W1) + b1[None, None, :, None, :])                # :63-69
    s = (torch.matmul(hid, W2) + b2[None, None, :, None, :]).transpose(1, 2).squeeze(-1)            # :70-75
    a = F.softmax(s, dim=-1)                                                         # :86
    o = torch.matmul(a, v)                                                           # :92
    return F.linear(o.permute(0, 2, 1, 3).reshape(b, m, E), w[p + ".out_proj.weight"])


def matnet_ffn(w: W, p: str, x: Tensor, x_old: Tensor) -> Tensor:
    """TransformerFFN.forward (in-tree copy attn_freenet.py:352-356) with FeedForward (:534-536)."""
    x = instance_norm(w, p + ".norm1", x_old + x)
    y = F.linear(F.relu(F.linear(x, w[p + ".ffn.W1.weight"], w[p + ".ffn.W1.bias"])), w[p + ".ffn.W2.weight"], w[p + ".ffn.W2.bias"])
    return instance_norm(w, p + ".norm2", x + y)


def matnet_encoder(w: W, td: dict, rand_idx: Tensor, layers: int, heads: int, env_name: str = "atsp", embed_dim: int = 256,
                   trace: Optional[dict] = None):
    """MatNetEncoder.forward (encoder.py:217-231), MatNetLayer (:168-172), MatNetMHA (:133-145); no attention mask
    (mask_non_neighbors=False)."""
    row, col, D = matnet_init_embedding(w, env_name, td, rand_idx, embed_dim)
    if trace is not None:
        trace["row0"], trace["col0"] = row, col
    for l in range(layers):
        p = f"layers.{l}"
        r_att = matnet_cross_mha(w, p + ".MHA.row_encoding_block", row, col, D, heads)
        c_att = matnet_cross_mha(w, p + ".MHA.col_encoding_block", col, row, D.transpose(-2, -1), heads)
        row, col = matnet_ffn(w, p + ".F_a.ops", r_att, row), matnet_ffn(w, p + ".F_b.ops", c_att, col)
        if trace is not None and l == 0:
            trace["row1"], trace["col1"] = row, col
    return row, col


def matnet_policy_template(embed_dim: int = 256, heads: int = 16, layers: int = 5, ff: int = 512, env_name: str = "atsp") -> Dict[str, tuple]:
    """state_dict of MatNetPolicy (rrnco/baselines/MatNet/policy.py:19-92): `encoder.*` + MatNetDecoder (decoder.py:24-87)."""
    E = embed_dim
    t = {"encoder." + k: v for k, v in matnet_weight_template(E, heads, layers, ff, env_name).items()}
    if env_name == "atsp":
        t["decoder.context_embedding.W_placeholder"] = (2 * E,)
        t["decoder.context_embedding.project_context.weight"] = (E, 2 * E)
    else:
        t["decoder.context_embedding.project_context.weight"] = (E, E + 1)
    t["decoder.pointer.project_out.weight"] = (E, E)
    t["decoder.project_node_embeddings.weight"] = (3 * E, E)
    t["decoder.project_fixed_context.weight"] = (E, E)
    return t


def matnet_pointer(w: W, q: Tensor, k: Tensor, v: Tensor, lk: Tensor, mask: Tensor, num_heads: int) -> Tensor:
    """rl4co PointerAttention [recalled; the in-tree RRNet_PointerAttention, rrnco/models/decoder.py:281-323, is this with the
    `project_out` step (still there, commented out, :295) replaced by a residual MLP]."""
    def heads(t):
        return t.unflatten(-1, (num_heads, -1)).transpose(-2, -3)
    am = mask.unsqueeze(1) if mask.ndim == 3 else mask.unsqueeze(1).unsqueeze(2)
    h = F.scaled_dot_product_attention(heads(q), heads(k), heads(v), attn_mask=am)
    g = F.linear(h.transpose(-2, -3).flatten(-2), w["decoder.pointer.project_out.weight"])
    return torch.bmm(g, lk.squeeze(-2).transpose(-2, -1)).squeeze(-2) / math.sqrt(g.size(-1))


def matnet_process_logits(logits: Tensor, mask: Tensor, temperature: float = 1.0, tanh_clipping: float = 10.0) -> Tensor:
    """process_logits of the MatNet baseline's own decoding module (rrnco/baselines/MatNet/decoding.py:316-372).  Unlike
    rrnco/models/decoding.py it shifts the row by its maximum and clamps to [-50, -1e-4] before the log-softmax (:357-359): every
    action within 1e-4 of the best one ties with it (argmax then takes the lowest index), and masked actions keep the finite
    logit -50 instead of -inf."""
    if tanh_clipping > 0:
        logits = torch.tanh(logits) * tanh_clipping
    logits = logits.clone()
    logits[~mask] = float("-inf")
    logits = logits.float() / temperature
    logits = logits - logits.max(dim=-1, keepdim=True).values
    logits = torch.clamp(logits, min=-50.0, max=-1e-4)
    return F.log_softmax(logits, dim=-1)


def matnet_policy_atsp(w: W, td0: dict, rand_idx: Tensor, num_starts: int, layers: int, heads: int, embed_dim: int = 256,
                       decode: str = "greedy", actions: Optional[Tensor] = None, trace: Optional[dict] = None) -> dict:
    """MatNetPolicy.forward (rrnco/baselines/MatNet/policy.py:94-212) for ATSP, use_graph_context=False, multistart greedy /
    evaluate: MatNetEncoder, MatNetDecoder._precompute_cache (decoder.py:89-113: glimpse key / value / logit key from the
    COLUMN embeddings, step context from the ROW embeddings), rl4co AttentionModelDecoder.forward [recalled; in-tree copy with
    an inductive bias: rrnco/models/decoder.py:151-206], the in-tree DecodingStrategy (MatNet/decoding.py), ATSPEnv."""
    we = {k[len("encoder."):]: v for k, v in w.items() if k.startswith("encoder.")}
    row, col = matnet_encoder(we, td0, rand_idx, layers, heads, "atsp", embed_dim)
    if trace is not None:
        trace["row_emb"], trace["col_emb"] = row, col
    B, N = td0["action_mask"].shape
    S = num_starts if num_starts > 1 else 0
    gk, gv, lk = F.linear(col, w["decoder.project_node_embeddings.weight"]).chunk(3, dim=-1)
    acts, lps = [], []
    if S >= 1:
        a0 = torch.arange(S).repeat_interleave(B) % N
        td = batchify_state({k: v for k, v in td0.items() if k not in ("locs",)}, S)
        td["action"] = a0
        td = atsp_step(td)
        lps.append(torch.zeros_like(a0, dtype=torch.float32)); acts.append(a0)
    else:
        td = dict(td0)
    k = 0
    while not td["done"].all():
        if S > 1:
            tv = {kk: unbatchify(td[kk], S) for kk in ("first_node", "current_node", "i", "action_mask")}
            tv["_two_d"] = True
        else:
            tv = {kk: td[kk] for kk in ("first_node", "current_node", "i", "action_mask")}
            tv["_two_d"] = False
        q = atsp_context(w, row, tv)                              # TSPContext on the row embeddings (+ graph context 0)
        q = q.unsqueeze(1) if q.ndim == 2 else q
        mask = tv["action_mask"]
        logits = matnet_pointer(w, q, gk, gv, lk, mask, heads)
        if S > 1:
            logits = logits.permute(1, 0, 2).reshape(-1, logits.shape[-1])
            mask = mask.permute(1, 0, 2).reshape(-1, mask.shape[-1])
        logp = matnet_process_logits(logits, mask)
        sel = logp.argmax(dim=-1) if decode == "greedy" else actions[:, k]
        if trace is not None:
            trace.setdefault("logits", []).append(logits); trace.setdefault("logp", []).append(logp)
        lps.append(gather_by_index(logp, sel, dim=1)); acts.append(sel)
        td["action"] = sel
        td = atsp_step(td)
        k += 1
    logprobs, actions_out = torch.stack(lps, 1), torch.stack(acts, 1)
    real, nd = atsp_reward(dict(td), actions_out, True)
    return {"reward": real, "normalized_reward": nd, "log_likelihood": logprobs.sum(1), "actions": actions_out, "logprobs": logprobs}


def matnet_policy_rcvrp(w: W, td0: dict, rand_idx: Tensor, num_starts: int, layers: int, heads: int, embed_dim: int = 256,
                        decode: str = "greedy", actions: Optional[Tensor] = None, trace: Optional[dict] = None) -> dict:
    """MatNetPolicy.forward for RCVRP — the environment configs/experiment/matnet.yaml trains on (RVRPInitEmbedding with
    use_coords=False, rl4co VRPContext: Linear(E+1, E) on [emb_cur; vehicle_capacity - used_capacity]); td0 = rcvrp_reset(...)."""
    we = {k[len("encoder."):]: v for k, v in w.items() if k.startswith("encoder.")}
    row, col = matnet_encoder(we, td0, rand_idx, layers, heads, "rcvrp", embed_dim)
    B, N1 = td0["action_mask"].shape
    n_loc = N1 - 1
    S = num_starts if num_starts > 1 else 0
    gk, gv, lk = F.linear(col, w["decoder.project_node_embeddings.weight"]).chunk(3, dim=-1)
    static = ("locs", "distance_matrix", "min_distance", "max_distance")
    acts, lps = [], []
    if S >= 1:
        a0 = torch.arange(S).repeat_interleave(B) % n_loc + 1
        td = batchify_state({k: v for k, v in td0.items() if k not in static}, S)
        td["action"] = a0
        td = rcvrp_step(td)
        lps.append(torch.zeros_like(a0, dtype=torch.float32)); acts.append(a0)
    else:
        td = {k: v for k, v in td0.items() if k not in static}
    k = 0
    while not td["done"].all():
        keys = ("current_node", "used_capacity", "vehicle_capacity", "action_mask")
        tv = {kk: (unbatchify(td[kk], S) if S > 1 else td[kk]) for kk in keys}
        cur = gather_by_index(row, tv["current_node"])
        q = F.linear(torch.cat([cur, tv["vehicle_capacity"] - tv["used_capacity"]], -1), w["decoder.context_embedding.project_context.weight"])
        q = q.unsqueeze(1) if q.ndim == 2 else q
        mask = tv["action_mask"]
        logits = matnet_pointer(w, q, gk, gv, lk, mask, heads)
        if S > 1:
            logits = logits.permute(1, 0, 2).reshape(-1, logits.shape[-1])
            mask = mask.permute(1, 0, 2).reshape(-1, mask.shape[-1])
        logp = matnet_process_logits(logits, mask)
        sel = logp.argmax(dim=-1) if decode == "greedy" else actions[:, k]
        if trace is not None:
            trace.setdefault("logits", []).append(logits); trace.setdefault("logp", []).append(logp)
        lps.append(gather_by_index(logp, sel, dim=1)); acts.append(sel)
        td["action"] = sel
        td = rcvrp_step(td)
        k += 1
    logprobs, actions_out = torch.stack(lps, 1), torch.stack(acts, 1)
    R = actions_out.shape[0]
    rtd = {"distance_matrix": td0["distance_matrix"][torch.arange(R) % B],
           "min_distance": td0["min_distance"][torch.arange(R) % B], "max_distance": td0["max_distance"][torch.arange(R) % B]}
    real, nd = vrp_reward(rtd, actions_out, True)
    return {"reward": real, "normalized_reward": nd, "log_likelihood": logprobs.sum(1), "actions": actions_out, "logprobs": logprobs}
