"""RRNetDecoder — drop-in for rrnco.models.decoder.RRNetDecoder (rrnco/models/decoder.py)."""
from __future__ import annotations

from dataclasses import dataclass

import torch
import torch.nn as nn

from .. import _lib as L


@dataclass
class PrecomputedCache:              # decoder.py:25-44 plus the per-node step-context tables
    node_embeddings: torch.Tensor
    graph_context: float
    glimpse_key: torch.Tensor        # [B,N,E]
    glimpse_val_t: torch.Tensor      # [B,E,112]  V^T, keys zero-padded (MFMA A operand of P.V)
    logit_key: torch.Tensor          # [B,N,E]
    ctx_a: torch.Tensor              # Wctx[:, :E] row_emb (ATSP first-node half) or None
    ctx_b: torch.Tensor              # Wctx[:, E:2E] row_emb (ATSP) / Wctx[:, :E] row_emb (VRP)
    split: tuple = None              # fp16 two-piece images of (glimpse_key, glimpse_val_t, logit_key) for the split rollout
    split_guarded: bool = False      # the images' range check has reported into a status word (rr_dec_cache / rr_pack_f16x2)

    def split_images(self, status=None):
        """K / V^T / L as two-piece fp16 images of 2^4 x (hi + lo, csrc/rr_common.h second form), built once per cache by
        rr_pack_f16x2; `status` (int32 device word): bit 0 is set when a value is non-finite or leaves the fp16 range."""
        if self.split is None or (status is not None and not self.split_guarded):      # (images built without a status word: check them now)
            out = []
            for t in (self.glimpse_key, self.glimpse_val_t, self.logit_key):
                d = torch.empty_like(t)
                L.check(L.lib().rr_pack_f16x2(L.ptr(t), L.ptr(d), t.numel(), L.ptr(status), L.stream()), "rr_pack_f16x2")
                out.append(d)
            self.split = tuple(out)
            self.split_guarded = status is not None
        return self.split

    @property
    def glimpse_val(self):
        n = self.glimpse_key.shape[1]
        return self.glimpse_val_t[:, :, :n].transpose(1, 2)


class _TSPContext(nn.Module):        # rl4co TSPContext: Linear(2E,E,bias=False) + W_placeholder[2E]
    def __init__(self, E):
        super().__init__()
        self.project_context = nn.Linear(2 * E, E, bias=False)
        self.W_placeholder = nn.Parameter(torch.Tensor(2 * E).uniform_(-1, 1))


class _VRPContext(nn.Module):        # rl4co VRPContext (E+1) / MTVRPContextEmbedding (E+4, context.py:34-70)
    def __init__(self, E, nstate):
        super().__init__()
        self.project_context = nn.Linear(E + nstate, E, bias=False)


class _MLP(nn.Module):               # rl4co MLP(E, E, [4E], ReLU) -> `lins.{0,1}`
    def __init__(self, E):
        super().__init__()
        self.lins = nn.ModuleList([nn.Linear(E, 4 * E), nn.Linear(4 * E, E)])


class RRNet_PointerAttention(nn.Module):   # decoder.py:235-279 (project_out exists but is unused, :271,295)
    def __init__(self, embed_dim, num_heads, out_bias=False, **unused):
        super().__init__()
        self.num_heads = num_heads
        self.project_out = nn.Linear(embed_dim, embed_dim, bias=out_bias)
        self.ffn = _MLP(embed_dim)


class RRNetDecoder(nn.Module):
    def __init__(self, embed_dim=128, num_heads=8, env_name="rcvrp", context_embedding=None, dynamic_embedding=None,
                 mask_inner=True, out_bias_pointer_attn=False, linear_bias=False, use_graph_context=True,
                 check_nan=True, sdpa_fn=None, pointer=None, moe_kwargs=None):
        super().__init__()
        if embed_dim != 128 or num_heads != 8 or linear_bias or not mask_inner:
            raise NotImplementedError("rrnco_amd decoder kernels: embed_dim=128, num_heads=8, no linear bias, mask_inner")
        self.env_name = getattr(env_name, "name", env_name)
        self.embed_dim, self.num_heads = embed_dim, num_heads
        if self.env_name == "rcvrptw":
            self.beta = nn.Parameter(torch.tensor([1.0]))
        E = embed_dim
        self.context_embedding = {"atsp": lambda: _TSPContext(E), "rcvrp": lambda: _VRPContext(E, 1),
                                  "rcvrptw": lambda: _VRPContext(E, 4)}[self.env_name]()
        self.pointer = RRNet_PointerAttention(E, num_heads, out_bias=out_bias_pointer_attn)
        self.project_node_embeddings = nn.Linear(E, 3 * E, bias=linear_bias)
        self.project_fixed_context = nn.Linear(E, E, bias=linear_bias)
        self.use_graph_context = use_graph_context
        self.alpha = nn.Parameter(torch.tensor([1.0]))

    def pre_decoder_hook(self, td, env, embeddings, num_starts: int = 0, packed=None, status=None):
        return td, env, self._precompute_cache(embeddings, num_starts, packed, status=status)

    def _precompute_cache(self, embeddings, num_starts: int = 0, packed=None, status=None) -> PrecomputedCache:
        """decoder.py:214-232 on csrc/rr_encoder.hip:k_dec_cache.  With the two-piece kernels on (the default) the same launch writes
        the rollout's fp16 images of K / V^T / L; `status` (the policy's range-guard word) collects their range check."""
        assert packed is not None
        row, col = (e.contiguous() for e in embeddings)
        L.require_gpu(row)
        Bp, N, E = row.shape
        if N > 103:
            from . import bign
            return bign.precompute_cache(self, row, col, packed)
        K, Lk, cb = torch.empty_like(row), torch.empty_like(row), torch.empty_like(row)
        ca = torch.empty_like(row) if self.env_name == "atsp" else None
        Vt = torch.empty(Bp, E, 112, device=row.device, dtype=torch.float32)
        from .. import packing
        from . import rollout as _R
        import os as _os
        images = None
        if _R.SPLIT_MLP and packing.mlp_split_enabled() and _os.environ.get("RR_CACHE_IMAGES", "1") != "0":     # (0: rr_pack_f16x2 on demand, A/B)
            images = (torch.empty_like(K), torch.empty_like(Vt), torch.empty_like(Lk))
        L.check(L.lib().rr_dec_cache(packed["cache"], L.ptr(row), L.ptr(col), L.ptr(K), L.ptr(Vt), L.ptr(Lk),
                                     L.ptr(ca), L.ptr(cb), *(L.ptr(t) for t in (images or (None, None, None))),
                                     L.ptr(status) if images is not None else None, Bp, N, L.stream()), "rr_dec_cache")
        return PrecomputedCache(row, 0, K, Vt, Lk, ca, cb, split=images, split_guarded=images is not None and status is not None)

    def forward(self, td, cached: PrecomputedCache, num_starts: int = 0, packed=None):
        """decoder.py:151-206: -> (logits [S*B,N] post inductive-bias transform, mask [S*B,N]).
        One launch of the rollout kernel in `logits_only` mode; the state is not modified."""
        from .rollout import launch_rollout
        R, N = td["action_mask"].shape
        if N > 103:
            from . import bign
            return bign.decoder_forward(self, td, cached, packed)
        logits = torch.empty(R, N, device=td.device, dtype=torch.float32)
        launch_rollout(self.env_name, packed, cached, td, num_starts, logits_out=logits, logits_only=True)
        return logits, td["action_mask"]
