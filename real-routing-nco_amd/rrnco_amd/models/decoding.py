"""Decoding strategies — drop-in for rrnco.models.decoding (greedy / sampling / evaluate, multistart).
logits -> log-probs -> action runs on csrc/rr_env.hip:k_select (decoding.py:311-361, 272-298, 266)."""
from __future__ import annotations

import torch

from .. import _lib as L
from ..ops import batchify


def get_decoding_strategy(decoding_strategy, **config):
    """rrnco/models/decoding.py:16-34."""
    registry = {"greedy": Greedy, "sampling": Sampling, "multistart_greedy": Greedy,
                "multistart_sampling": Sampling, "evaluate": Evaluate, "beam_search": BeamSearch}
    if "multistart" in decoding_strategy:
        config["multistart"] = True
    return registry.get(decoding_strategy, Sampling)(**config)


class DecodingStrategy:
    name = "base"
    mode = "greedy"

    def __init__(self, temperature=1.0, top_p=0.0, top_k=0, mask_logits=True, tanh_clipping=0, num_samples=None,
                 multisample=False, num_starts=None, multistart=False, select_start_nodes_fn=None,
                 improvement_method_mode=False, select_best=False, store_all_logp=False, seed=None, **kwargs):
        if improvement_method_mode:
            raise NotImplementedError("improvement_method_mode is outside the MI355X hot path")
        self.select_best = bool(select_best)
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
        # the sampling noise is a pure function of (seed, rollout, step, key): without an explicit seed, draw a fresh one from
        # torch's generator, as the reference's torch.multinomial advances it (decoding.py:282-298)
        self.seed = int(torch.randint(0, 2 ** 62, (1,)).item()) if seed is None else int(seed)
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
        logprobs, actions = torch.stack(self.logprobs, 1), torch.stack(self.actions, 1)
        if self.num_starts > 0 and self.select_best:                                   # decoding.py:214-216
            logprobs, actions, td, env = self._select_best(logprobs, actions, td, env)
        return logprobs, actions, td, env

    def _select_best(self, logprobs, actions, td, env):
        """decoding.py:300-309: keep, per instance, the start / sample with the best reward.  (The reference takes `.max` of
        env.get_reward's return value, which is a (real, normalised) tuple for these envs: the real reward is used here.)"""
        rew = env.get_reward(td, actions)
        rew = rew[0] if isinstance(rew, tuple) else rew
        B = rew.shape[0] // self.num_starts
        best = rew.view(self.num_starts, B).argmax(0)                                  # unbatchify(rew, S).max(-1)
        rows = best * B + torch.arange(B, device=rew.device)                           # r = s * B + b
        return logprobs[rows], actions[rows], td.index_rollouts(rows), env

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
        if N > 128 and not getattr(self, "matnet_clamp", False):       # rows of up to 1 024 keys (csrc/rr_bign.hip)
            L.check(L.lib().rr_select_big(L.ptr(logits), L.ptr(m), L.ptr(act_in), L.ptr(sel), L.ptr(lp), L.ptr(lp_all), R, N,
                                          float(self.tanh_clipping), float(self.temperature), mode, int(self.seed),
                                          len(self.actions), self.top_k, self.top_p, L.stream()), "rr_select_big")
        elif getattr(self, "matnet_clamp", False):       # the MatNet baseline's own process_logits (MatNet/decoding.py:316-372)
            L.check(L.lib().rr_select_matnet(L.ptr(logits), L.ptr(m), L.ptr(act_in), L.ptr(sel), L.ptr(lp), L.ptr(lp_all), R, N,
                                             float(self.tanh_clipping), float(self.temperature), mode, int(self.seed),
                                             len(self.actions), L.stream()), "rr_select_matnet")
        else:
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


class BeamSearch(DecodingStrategy):
    """rrnco/models/decoding.py:402-554.  The per-step log-probabilities come from rr_select (all-logp output); the beam
    bookkeeping (top-k over the stacked beams, parent pointers, back-tracking, best-beam selection) is index arithmetic on
    [B, W] tensors and stays in torch ops on the device.  Runs in the step-wise loop (the fused rollout has no beams)."""
    name, mode = "beam_search", "greedy"

    def __init__(self, beam_width=None, select_best=True, **kwargs):
        kwargs["store_all_logp"] = True
        kwargs.pop("select_best", None)
        super().__init__(**kwargs)
        self.beam_width, self.select_best_beam = beam_width, select_best
        self.parent_beam_logprobs, self.beam_path = None, []
        self.is_beam_search = True

    def pre_decoder_hook(self, td, env, action=None):
        """:429-454."""
        if self.beam_width is None:
            self.beam_width = env.get_num_starts(td)
        assert self.beam_width > 1, "beam width must be larger than 1"
        action = (self.select_start_nodes_fn(td, env, self.beam_width) if self.select_start_nodes_fn
                  else env.select_start_nodes(td, num_starts=self.beam_width))
        td = batchify(td, self.beam_width)
        td.set("action", action)
        td = env.step(td)["next"]
        logprobs = torch.zeros_like(td["action_mask"], dtype=torch.float32)
        self.logprobs.append(logprobs)
        self.actions.append(action)
        self.parent_beam_logprobs = logprobs.gather(1, action[..., None])
        self.beam_path.append(torch.zeros(logprobs.size(0), device=td.device, dtype=torch.int32))
        self.num_starts = self.beam_width
        return td, env, self.beam_width

    def step(self, logits, mask, td=None, action=None, **kwargs):
        """:219-270 with BeamSearch._step (:414-427) and _make_beam_step (:507-554)."""
        logits = logits.contiguous()
        R, N = logits.shape
        m = mask.contiguous() if (self.mask_logits and mask is not None) else None
        sel = torch.empty(R, dtype=torch.int64, device=logits.device)
        lp = torch.empty(R, dtype=torch.float32, device=logits.device)
        lp_all = torch.empty(R, N, dtype=torch.float32, device=logits.device)
        L.check(L.lib().rr_select(L.ptr(logits), L.ptr(m), None, L.ptr(sel), L.ptr(lp), L.ptr(lp_all), R, N,
                                  float(self.tanh_clipping), float(self.temperature), 0, 0, len(self.actions),
                                  self.top_k, self.top_p, L.stream()), "rr_select")
        B = R // self.beam_width
        seq = torch.arange(B, device=logits.device).repeat(self.beam_width)
        stacked = torch.cat((lp_all + self.parent_beam_logprobs).split(B), dim=1)           # [B, W*N]
        top_lp, top_ix = torch.topk(stacked, self.beam_width, dim=1)
        self.parent_beam_logprobs = torch.hstack(torch.unbind(top_lp, 1)).unsqueeze(1)
        top_ix = torch.hstack(torch.unbind(top_ix, 1))
        selected = top_ix % N
        parent = (top_ix // N).int()
        idx = seq + parent * B
        self.beam_path.append(parent)
        td = td.index_rollouts(idx)
        lp_all = lp_all[idx]
        assert not (~mask[idx]).gather(1, selected.unsqueeze(-1)).any(), "infeasible action selected"
        td.set("action", selected)
        self.actions.append(selected)
        self.logprobs.append(lp_all)
        return td

    def post_decoder_hook(self, td, env):
        """:456-505."""
        actions, logprobs = torch.stack(self.actions, 1), torch.stack(self.logprobs, 1)
        assert actions.size(1) == len(self.beam_path), "action idx shape and beam path shape dont match"
        cur = self.beam_path[-1]
        seqs, lps = [actions[:, -1]], [logprobs[:, -1]]
        R = actions.size(0)
        B = R // self.beam_width
        seq = torch.arange(B, device=actions.device).repeat(self.beam_width)
        for k in reversed(range(len(self.beam_path) - 1)):
            idx = seq + cur * B
            seqs.append(actions[idx, k]); lps.append(logprobs[idx, k])
            cur = self.beam_path[k][idx]
        actions, logprobs = torch.stack(list(reversed(seqs)), 1), torch.stack(list(reversed(lps)), 1)
        if not self.select_best_beam:
            return logprobs, actions, td, env
        rewards = env.get_reward(td, actions)
        rewards = rewards[0] if isinstance(rewards, tuple) else rewards
        _, best = torch.cat(rewards.unsqueeze(1).split(B), 1).max(1)
        flat = torch.arange(B, device=rewards.device) + best * B
        return logprobs[flat], actions[flat], td.index_rollouts(flat), env
