"""Decoding strategies — drop-in for rrnco.models.decoding (greedy / sampling / evaluate, multistart).
logits -> log-probs -> action runs on csrc/rr_env.hip:k_select (decoding.py:311-361, 272-298, 266)."""
from __future__ import annotations

import torch

from .. import _lib as L
from ..ops import batchify


def get_decoding_strategy(decoding_strategy, **config):
    """rrnco/models/decoding.py:16-34 (beam_search is out of scope: unused by the reference configs)."""
    registry = {"greedy": Greedy, "sampling": Sampling, "multistart_greedy": Greedy,
                "multistart_sampling": Sampling, "evaluate": Evaluate}
    if decoding_strategy == "beam_search":
        raise NotImplementedError("beam_search is not part of the MI355X hot path")
    if "multistart" in decoding_strategy:
        config["multistart"] = True
    return registry.get(decoding_strategy, Sampling)(**config)


class DecodingStrategy:
    name = "base"
    mode = "greedy"

    def __init__(self, temperature=1.0, top_p=0.0, top_k=0, mask_logits=True, tanh_clipping=0, num_samples=None,
                 multisample=False, num_starts=None, multistart=False, select_start_nodes_fn=None,
                 improvement_method_mode=False, select_best=False, store_all_logp=False, seed=0, **kwargs):
        if select_best or improvement_method_mode:
            raise NotImplementedError("select_best / improvement_method_mode are outside the MI355X hot path")
        if not 0.0 <= top_p <= 1.0:
            raise AssertionError("top-p should be in (0, 1].")                      # decoding.py:357
        # decoding.py:352-358 filters; evaluated by rr_select in the step-wise loop (the fused rollout has no sort)
        self.top_p, self.top_k = float(top_p), int(top_k)
        self.temperature, self.mask_logits, self.tanh_clipping = temperature, mask_logits, tanh_clipping
        assert not (multistart and multisample)
        if num_samples is not None:
            multisample = num_samples > 1
        if num_starts is not None:
            multistart = num_starts > 1
        self.multistart, self.multisample = multistart, multisample
        self.num_starts = num_starts if multistart else num_samples
        self.select_start_nodes_fn = select_start_nodes_fn
        self.store_all_logp = store_all_logp
        self.seed = seed
        self.actions, self.logprobs = [], []

    def pre_decoder_hook(self, td, env, action=None):
        """decoding.py:157-205."""
        if self.multistart or self.multisample:
            if self.num_starts is None:
                self.num_starts = env.get_num_starts(td)
        else:
            self.num_starts = 0
        if self.num_starts >= 1:
            if self.multistart:
                if action is None:
                    action = (self.select_start_nodes_fn(td, env, self.num_starts) if self.select_start_nodes_fn
                              else env.select_start_nodes(td, num_starts=self.num_starts))
                td = batchify(td, self.num_starts)
                td.set("action", action)
                td = env.step(td)["next"]
                lp = torch.zeros_like(td["action_mask"], dtype=torch.float32) if self.store_all_logp else \
                    torch.zeros(action.shape, device=td.device, dtype=torch.float32)
                self.logprobs.append(lp)
                self.actions.append(action)
            else:
                td = batchify(td, self.num_starts)
        return td, env, self.num_starts

    def post_decoder_hook(self, td, env):
        assert len(self.logprobs) > 0, "No logprobs were collected because all environments were done"
        return torch.stack(self.logprobs, 1), torch.stack(self.actions, 1), td, env

    def step(self, logits, mask, td=None, action=None, **kwargs):
        """decoding.py:219-270."""
        logits = logits.contiguous()
        R, N = logits.shape
        m = mask.contiguous() if (self.mask_logits and mask is not None) else None
        sel = torch.empty(R, dtype=torch.int64, device=logits.device)
        lp = torch.empty(R, dtype=torch.float32, device=logits.device)
        lp_all = torch.empty(R, N, dtype=torch.float32, device=logits.device) if self.store_all_logp else None
        mode = {"greedy": 0, "sampling": 1, "evaluate": 2}[self.mode]
        act_in = action.contiguous() if action is not None else None
        L.check(L.lib().rr_select(L.ptr(logits), L.ptr(m), L.ptr(act_in), L.ptr(sel), L.ptr(lp), L.ptr(lp_all), R, N,
                                  float(self.tanh_clipping), float(self.temperature), mode, int(self.seed),
                                  len(self.actions), self.top_k, self.top_p, L.stream()), "rr_select")
        td.set("action", sel)
        self.actions.append(sel)
        self.logprobs.append(lp_all if self.store_all_logp else lp)
        return td


class Greedy(DecodingStrategy):
    name, mode = "greedy", "greedy"


class Sampling(DecodingStrategy):
    name, mode = "sampling", "sampling"


class Evaluate(DecodingStrategy):
    name, mode = "evaluate", "evaluate"
