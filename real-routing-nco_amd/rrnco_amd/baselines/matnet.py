"""MatNet baseline encoder — drop-in for rrnco.baselines.MatNet.encoder.MatNetEncoder (encoder.py:175-231) on the HIP
kernels of csrc/rr_matnet.hip.  Parameter names and shapes are the reference's (`layers.{l}.MHA.{row,col}_encoding_block.
{Wq,Wkv,out_proj}.weight`, `.sdpa_fn.mix_{W1,b1,W2,b2}`, `layers.{l}.F_{a,b}.ops.{norm1,norm2}.normalizer.*`,
`.ops.ffn.{W1,W2}.*`, `init_embedding.*`), so a reference state_dict loads with strict=True.

Scope: the encoder.  The reference's MatNet decoder is rl4co's AttentionModelDecoder at 256 / 16 heads;
it is not part of this module."""
from __future__ import annotations

import ctypes as C

import torch
import torch.nn as nn

from .. import _lib as L
from ..packing import pack_a, weights_fingerprint


class _MixedScoresSDPA(nn.Module):
    """encoder.py:14-43 (parameters only; the arithmetic runs in k_mn_attn)."""

    def __init__(self, num_heads, mixer_hidden_dim=16, mix1_init=(1 / 2) ** (1 / 2), mix2_init=(1 / 16) ** (1 / 2)):
        super().__init__()
        u = lambda b, *s: nn.Parameter(torch.empty(*s).uniform_(-b, b))  # noqa: E731
        self.mix_W1, self.mix_b1 = u(mix1_init, num_heads, 2, mixer_hidden_dim), u(mix1_init, num_heads, mixer_hidden_dim)
        self.mix_W2, self.mix_b2 = u(mix2_init, num_heads, mixer_hidden_dim, 1), u(mix2_init, num_heads, 1)


class _CrossMHA(nn.Module):
    """MatNetCrossMHA (encoder.py:95-116) = rl4co MultiHeadCrossAttention(Wq, Wkv, out_proj) around MixedScoresSDPA."""

    def __init__(self, embed_dim, num_heads, bias=False):
        super().__init__()
        self.sdpa_fn = _MixedScoresSDPA(num_heads)
        self.Wq = nn.Linear(embed_dim, embed_dim, bias=bias)
        self.Wkv = nn.Linear(embed_dim, 2 * embed_dim, bias=bias)
        self.out_proj = nn.Linear(embed_dim, embed_dim, bias=bias)


class _MHA(nn.Module):
    def __init__(self, embed_dim, num_heads, bias=False):
        super().__init__()
        self.row_encoding_block = _CrossMHA(embed_dim, num_heads, bias)
        self.col_encoding_block = _CrossMHA(embed_dim, num_heads, bias)


class _Norm(nn.Module):
    def __init__(self, embed_dim):
        super().__init__()
        self.normalizer = nn.InstanceNorm1d(embed_dim, affine=True)


class _FeedForward(nn.Module):
    def __init__(self, embed_dim, hidden):
        super().__init__()
        self.W1, self.W2 = nn.Linear(embed_dim, hidden), nn.Linear(hidden, embed_dim)


class _TransformerFFN(nn.Module):
    def __init__(self, embed_dim, hidden):
        super().__init__()
        self.ops = nn.ModuleDict({"norm1": _Norm(embed_dim), "ffn": _FeedForward(embed_dim, hidden), "norm2": _Norm(embed_dim)})


class _Layer(nn.Module):
    def __init__(self, embed_dim, num_heads, bias, hidden):
        super().__init__()
        self.MHA = _MHA(embed_dim, num_heads, bias)
        self.F_a, self.F_b = _TransformerFFN(embed_dim, hidden), _TransformerFFN(embed_dim, hidden)


class _ATSPInit(nn.Module):
    """env_embeddings/atsp.py: no parameters."""

    def __init__(self, embed_dim, **unused):
        super().__init__()


class _RVRPInit(nn.Module):
    """env_embeddings/rcvrp.py:14-35 with use_coords=False (configs/experiment/matnet.yaml:27-29)."""

    def __init__(self, embed_dim, linear_bias=True, use_coords=False, use_polar_feats=False):
        super().__init__()
        if use_coords or use_polar_feats:
            raise NotImplementedError("MatNet RVRPInitEmbedding with use_coords / use_polar_feats")
        self.depot_client_emb = nn.Embedding(2, embed_dim)
        self.init_embed = nn.Linear(1, embed_dim, linear_bias)
        self.row_combine_embed = nn.Linear(embed_dim * 2, embed_dim, linear_bias)
        self.col_combine_embed = nn.Linear(embed_dim * 2, embed_dim, linear_bias)


class MatNetEncoder(nn.Module):
    def __init__(self, embed_dim: int = 256, num_heads: int = 16, num_layers: int = 5, normalization: str = "instance",
                 feedforward_hidden: int = 512, init_embedding: nn.Module = None, env_name: str = "rcvrp",
                 init_embedding_kwargs: dict = {}, bias: bool = False, mask_non_neighbors: bool = False):
        super().__init__()
        env_name = getattr(env_name, "name", env_name)
        if normalization != "instance" or bias or mask_non_neighbors or init_embedding is not None:
            raise NotImplementedError("MatNetEncoder on HIP: instance normalisation, no projection biases, no attention mask")
        if embed_dim != 16 * num_heads or embed_dim % 256 or feedforward_hidden % 256:
            raise NotImplementedError("MatNetEncoder on HIP: head dim 16, embed_dim and feedforward_hidden multiples of 256")
        if env_name not in ("atsp", "rcvrp"):
            raise ValueError(f"Unknown environment name '{env_name}'")          # env_embeddings/__init__.py:20-23
        self.env_name, self.embed_dim, self.num_heads, self.ff = env_name, embed_dim, num_heads, feedforward_hidden
        self.init_embedding = (_ATSPInit if env_name == "atsp" else _RVRPInit)(embed_dim, **init_embedding_kwargs)
        self.layers = nn.ModuleList([_Layer(embed_dim, num_heads, bias, feedforward_hidden) for _ in range(num_layers)])
        self._pack_cache = None

    # ---- MFMA-ordered weights, rebuilt when any parameter changes
    def packed(self, device):
        key = (str(device), tuple(p._version for p in self.parameters()), tuple(p.data_ptr() for p in self.parameters()),
               weights_fingerprint(self))
        if self._pack_cache is None or self._pack_cache[0] != key:
            self._pack_cache = (key, self._pack(device))
        return self._pack_cache[1]

    def _pack(self, device):
        keep = []

        def put(t):
            t = t.detach().to(device=device, dtype=torch.float32).contiguous()
            keep.append(t)
            return t.data_ptr()

        sd = {k: v.detach().double().cpu() for k, v in self.state_dict().items()}
        E, H = self.embed_dim, self.num_heads
        layers = []
        for l in range(len(self.layers)):
            sides = []
            for blk, ffn in (("row_encoding_block", "F_a"), ("col_encoding_block", "F_b")):
                p, f = f"layers.{l}.MHA.{blk}", f"layers.{l}.{ffn}.ops"
                W1m, b1m = sd[p + ".sdpa_fn.mix_W1"], sd[p + ".sdpa_fn.mix_b1"]
                mix = torch.zeros(H, 68, dtype=torch.float64)
                mix[:, 0:16] = W1m[:, 0, :] / 4.0                   # 1 / sqrt(head dim 16) folded into the score row (exact)
                mix[:, 16:32], mix[:, 32:48] = W1m[:, 1, :], b1m
                mix[:, 48:64], mix[:, 64] = sd[p + ".sdpa_fn.mix_W2"][:, :, 0], sd[p + ".sdpa_fn.mix_b2"][:, 0]
                w = L.MatNetSideW()
                w.wq, w.wkv, w.wo = put(pack_a(sd[p + ".Wq.weight"].float())), put(pack_a(sd[p + ".Wkv.weight"].float())), \
                    put(pack_a(sd[p + ".out_proj.weight"].float()))
                w.w1, w.w2 = put(pack_a(sd[f + ".ffn.W1.weight"].float())), put(pack_a(sd[f + ".ffn.W2.weight"].float()))
                w.b1, w.b2 = put(sd[f + ".ffn.W1.bias"]), put(sd[f + ".ffn.W2.bias"])
                w.n1g, w.n1b = put(sd[f + ".norm1.normalizer.weight"]), put(sd[f + ".norm1.normalizer.bias"])
                w.n2g, w.n2b = put(sd[f + ".norm2.normalizer.weight"]), put(sd[f + ".norm2.normalizer.bias"])
                w.mix = put(mix)
                sides.append(w)
            layers.append(tuple(sides))
        init = None
        if self.env_name == "rcvrp":
            # rcvrp.py:70-81 folded (float64): out = emb0 (depot) | emb1 + w_i demand + b_i; row / col = W[:, E:] out + b
            q = "init_embedding"
            emb, wi, bi = sd[q + ".depot_client_emb.weight"], sd[q + ".init_embed.weight"][:, 0], sd[q + ".init_embed.bias"]
            vecs = []
            for nm in ("row_combine_embed", "col_combine_embed"):
                W2, bb = sd[f"{q}.{nm}.weight"][:, E:], sd[f"{q}.{nm}.bias"]
                vecs.append(torch.stack([W2 @ emb[0] + bb, W2 @ (emb[1] + bi) + bb, W2 @ wi]))
            init = (put(vecs[0]), put(vecs[1]), put(sd[q + ".col_combine_embed.weight"][:, :E].t()))
        return {"layers": layers, "init": init, "keep": keep}

    @torch.no_grad()
    def forward(self, td, attn_mask: torch.Tensor = None, rand_idx: torch.Tensor = None):
        """-> ((row_emb, col_emb), None), encoder.py:217-231.  `rand_idx` [B, N] pins the random one-hot slots of the column
        embeddings (the reference draws `torch.rand(b, c).argsort(1)` on every call); also read from td["rand_idx"]."""
        if attn_mask is not None:
            raise NotImplementedError("MatNetEncoder on HIP: attn_mask")
        D = td["distance_matrix"]
        L.require_gpu(D)
        D = D.float().contiguous()
        Bp, N, _ = D.shape
        E = self.embed_dim
        if N > E:
            raise ValueError("MatNet one-hot column embedding needs N <= embed_dim")
        dev = D.device
        pk = self.packed(dev)
        if rand_idx is None:
            rand_idx = td.get("rand_idx", None)
        if rand_idx is None:
            rand_idx = torch.rand(Bp, N, device=dev).argsort(dim=1)            # env_embeddings/atsp.py:29-30
        rand_idx = rand_idx.to(device=dev, dtype=torch.int64).contiguous()
        lib = L.lib()
        row, col = torch.empty(Bp, N, E, device=dev), torch.empty(Bp, N, E, device=dev)
        row2, col2 = torch.empty_like(row), torch.empty_like(col)
        if self.env_name == "atsp":
            L.check(lib.rr_matnet_init(L.ptr(rand_idx), None, None, None, None, L.ptr(row), L.ptr(col), Bp, N, E, L.stream()), "rr_matnet_init")
        else:
            dem = td["demand"].float().contiguous()
            rv, cv, st = pk["init"]
            L.check(lib.rr_matnet_init(L.ptr(rand_idx), L.ptr(dem), rv, cv, st, L.ptr(row), L.ptr(col), Bp, N, E, L.stream()), "rr_matnet_init")
        nbytes = lib.rr_matnet_workspace_bytes(Bp, N, E, self.ff)
        ws = torch.empty(nbytes // 4, device=dev, dtype=torch.float32)
        for wr, wc in pk["layers"]:
            L.check(lib.rr_matnet_layer(C.byref(wr), C.byref(wc), L.ptr(row), L.ptr(col), L.ptr(row2), L.ptr(col2), L.ptr(D),
                                        L.ptr(ws), nbytes, Bp, N, E, self.num_heads, self.ff, L.stream()), "rr_matnet_layer")
            row, col, row2, col2 = row2, col2, row, col
        return (row, col), None


# ----------------------------------------------------------------------------------------------------------------------
# Decoder and policy: rrnco/baselines/MatNet/decoder.py (rl4co AttentionModelDecoder + PointerAttention) and policy.py
# ----------------------------------------------------------------------------------------------------------------------
class _ProjectContext(nn.Module):
    """rl4co TSPContext: `project_context` Linear(2E, E, bias=False) on [emb_first; emb_cur], `W_placeholder` before the first move."""

    def __init__(self, embed_dim):
        super().__init__()
        self.W_placeholder = nn.Parameter(torch.empty(2 * embed_dim).uniform_(-1, 1))
        self.project_context = nn.Linear(2 * embed_dim, embed_dim, bias=False)


class _ProjectContextVRP(nn.Module):
    """rl4co VRPContext: `project_context` Linear(E + 1, E, bias=False) on [emb_cur; vehicle_capacity - used_capacity]."""

    def __init__(self, embed_dim):
        super().__init__()
        self.project_context = nn.Linear(embed_dim + 1, embed_dim, bias=False)


class _Pointer(nn.Module):
    def __init__(self, embed_dim):
        super().__init__()
        self.project_out = nn.Linear(embed_dim, embed_dim, bias=False)


class MatNetDecoder(nn.Module):
    """decoder.py:24-113 for ATSP: cache (glimpse key / value / logit key from the COLUMN embeddings, step context from the ROW
    embeddings, no graph context) by rr_matnet_linear, one forward per decode step by rr_matnet_dec_step."""

    def __init__(self, embed_dim: int = 256, num_heads: int = 16, env_name: str = "atsp", use_graph_context: bool = False, **unused):
        super().__init__()
        if env_name not in ("atsp", "rcvrp") or use_graph_context or embed_dim != 256 or num_heads != 16:
            raise NotImplementedError("MatNetDecoder on HIP: ATSP / RCVRP, embed_dim=256, num_heads=16, use_graph_context=False "
                                      "(configs/experiment/matnet.yaml)")
        self.embed_dim, self.num_heads, self.env_name = embed_dim, num_heads, env_name
        self.context_embedding = _ProjectContext(embed_dim) if env_name == "atsp" else _ProjectContextVRP(embed_dim)
        self.pointer = _Pointer(embed_dim)
        self.project_node_embeddings = nn.Linear(embed_dim, 3 * embed_dim, bias=False)
        self.project_fixed_context = nn.Linear(embed_dim, embed_dim, bias=False)      # unused without the graph context (as in the reference)
        self._pack_cache = None

    def packed(self, device):
        key = (str(device), tuple(p._version for p in self.parameters()), tuple(p.data_ptr() for p in self.parameters()),
               weights_fingerprint(self))
        if self._pack_cache is None or self._pack_cache[0] != key:
            E = self.embed_dim
            f = lambda t: pack_a(t.detach().float()).to(device).contiguous()          # noqa: E731
            Wc = self.context_embedding.project_context.weight
            pk = {"wnode": f(self.project_node_embeddings.weight), "wo": f(self.pointer.project_out.weight)}
            if self.env_name == "atsp":
                pk["wca"], pk["wcb"] = f(Wc[:, :E]), f(Wc[:, E:])
                pk["q0"] = (Wc.detach().double() @ self.context_embedding.W_placeholder.detach().double()).float().to(device).contiguous()
            else:
                pk["wcb"], pk["wstate"] = f(Wc[:, :E]), Wc[:, E].detach().float().to(device).contiguous()
            self._pack_cache = (key, pk)
        return self._pack_cache[1]

    @torch.no_grad()
    def pre_decoder_hook(self, td, env, embeddings, num_starts: int = 0):
        """-> td, env, cache (decoder.py:89-113 `_precompute_cache`)."""
        row, col = embeddings
        Bp, N, E = row.shape
        pk, lib = self.packed(row.device), L.lib()
        kvl = torch.empty(Bp, N, 3 * E, device=row.device)
        ctxa, ctxb = None, torch.empty(Bp, N, E, device=row.device)
        row, col = row.contiguous(), col.contiguous()
        L.check(lib.rr_matnet_linear(L.ptr(pk["wnode"]), L.ptr(col), L.ptr(kvl), Bp, N, E, 3 * E, L.stream()), "rr_matnet_linear")
        if self.env_name == "atsp":
            ctxa = torch.empty(Bp, N, E, device=row.device)
            L.check(lib.rr_matnet_linear(L.ptr(pk["wca"]), L.ptr(row), L.ptr(ctxa), Bp, N, E, E, L.stream()), "rr_matnet_linear")
        L.check(lib.rr_matnet_linear(L.ptr(pk["wcb"]), L.ptr(row), L.ptr(ctxb), Bp, N, E, E, L.stream()), "rr_matnet_linear")
        vt = torch.zeros(Bp, E, 112, device=row.device)                           # V^T, keys along the row (16-byte operand loads)
        vt[:, :, :N] = kvl[:, :, E:2 * E].transpose(1, 2)
        return td, env, {"kvl": kvl, "vt": vt, "ctxA": ctxa, "ctxB": ctxb, "Bp": Bp, "N": N}

    @torch.no_grad()
    def forward(self, td, cached, num_starts: int = 0):
        """-> logits [R, N], mask [R, N] (AttentionModelDecoder.forward; r = s * B + b)."""
        mask = td["action_mask"].contiguous()
        R, N = mask.shape
        Bp = cached["Bp"]
        S = R // Bp
        pk = self.packed(mask.device)
        logits = torch.empty(R, N, device=mask.device)
        state = wstate = q0 = first = None
        if self.env_name == "atsp":
            placeholder = td.meta.get("i", 1) == 0                # nothing visited yet (plain greedy / sampling, first step)
            q0 = pk["q0"]
            first = None if placeholder else td["first_node"].reshape(-1).contiguous()
            cur = None if placeholder else td["current_node"].reshape(-1).contiguous()
        else:                                                     # VRPContext: [emb_cur; vehicle_capacity - used_capacity]
            cur = td["current_node"].reshape(-1).contiguous()
            vcap = td["vehicle_capacity"].reshape(-1)
            state = ((vcap if vcap.shape[0] == R else vcap.repeat(R // vcap.shape[0])) - td["used_capacity"].reshape(-1)).float().contiguous()
            wstate = pk["wstate"]
        L.check(L.lib().rr_matnet_dec_step(L.ptr(pk["wo"]), L.ptr(cached["kvl"]), L.ptr(cached["vt"]), L.ptr(cached["ctxA"]), L.ptr(cached["ctxB"]),
                                           L.ptr(q0), L.ptr(state), L.ptr(wstate), L.ptr(first), L.ptr(cur), L.ptr(mask.view(torch.uint8)),
                                           L.ptr(logits), Bp, N, S, self.embed_dim, self.num_heads, L.stream()), "rr_matnet_dec_step")
        return logits, mask


class MatNetPolicy(nn.Module):
    """rrnco.baselines.MatNet.policy.MatNetPolicy (policy.py:19-212) for ATSP and RCVRP (the environment matnet.yaml trains on): MatNetEncoder -> MatNetDecoder, decoded with the
    baseline's own process_logits (MatNet/decoding.py: row shifted by its maximum and clamped to [-50, -1e-4], rr_select_matnet)
    in the reference's step loop (decoder.forward -> strategy.step -> env.step)."""

    def __init__(self, env_name: str = "atsp", embed_dim: int = 256, num_encoder_layers: int = 5, num_heads: int = 16,
                 normalization: str = "instance", use_graph_context: bool = False, temperature: float = 1.0,
                 tanh_clipping: float = 10.0, mask_logits: bool = True, bias: bool = False, init_embedding_kwargs: dict = {},
                 train_decode_type: str = "sampling", val_decode_type: str = "greedy", test_decode_type: str = "greedy", **unused):
        super().__init__()
        self.env_name = getattr(env_name, "name", env_name)
        self.encoder = MatNetEncoder(embed_dim=embed_dim, num_heads=num_heads, num_layers=num_encoder_layers, normalization=normalization,
                                     env_name=self.env_name, init_embedding_kwargs=init_embedding_kwargs, bias=bias)
        self.decoder = MatNetDecoder(embed_dim=embed_dim, num_heads=num_heads, env_name=self.env_name, use_graph_context=use_graph_context)
        self.temperature, self.tanh_clipping, self.mask_logits = temperature, tanh_clipping, mask_logits
        self.train_decode_type, self.val_decode_type, self.test_decode_type = train_decode_type, val_decode_type, test_decode_type

    @torch.no_grad()
    def forward(self, td, env=None, phase: str = "train", calc_reward: bool = True, return_actions: bool = True,
                return_hidden: bool = False, return_sum_log_likelihood: bool = True, actions=None, max_steps: int = 1_000_000,
                rand_idx=None, **decoding_kwargs) -> dict:
        from ..models.decoding import get_decoding_strategy
        from ..ops import get_log_likelihood
        if env is None or isinstance(env, str):
            raise ValueError("pass an instantiated rrnco_amd env")
        hidden, _ = self.encoder(td, rand_idx=rand_idx)
        decode_type = decoding_kwargs.pop("decode_type", None)
        if actions is not None:
            decode_type = "evaluate"
        elif decode_type is None:
            decode_type = getattr(self, f"{phase}_decode_type")
        strategy = get_decoding_strategy(decode_type, temperature=decoding_kwargs.pop("temperature", self.temperature),
                                         tanh_clipping=decoding_kwargs.pop("tanh_clipping", self.tanh_clipping),
                                         mask_logits=decoding_kwargs.pop("mask_logits", self.mask_logits), **decoding_kwargs)
        if strategy.top_k or strategy.top_p or getattr(strategy, "is_beam_search", False) or strategy.store_all_logp:
            raise NotImplementedError("MatNetPolicy on HIP: greedy / sampling / evaluate without top-k / top-p filters")
        strategy.matnet_clamp = True                              # MatNet/decoding.py:357-359 inside rr_select_matnet
        td, env, num_starts = strategy.pre_decoder_hook(td, env)
        td, env, cache = self.decoder.pre_decoder_hook(td, env, hidden, num_starts)
        step = 0
        while not td["done"].all():
            logits, mask = self.decoder(td, cache, num_starts)
            td = strategy.step(logits, mask, td, action=actions[..., step] if actions is not None else None)
            td = env.step(td)["next"]
            step += 1
            if step > max_steps:
                break
        logprobs, actions_out, td, env = strategy.post_decoder_hook(td, env)
        out = {}
        if calc_reward:
            if env.normalize:
                real, normd = env.get_reward(td, actions_out)
                out["reward"], out["normalized_reward"] = real, normd
            else:
                out["reward"] = env.get_reward(td, actions_out)
        out["log_likelihood"] = get_log_likelihood(logprobs, actions_out, td.get("mask", None), return_sum_log_likelihood)
        if return_actions:
            out["actions"] = actions_out
        if return_hidden:
            out["hidden"] = hidden
        return out
