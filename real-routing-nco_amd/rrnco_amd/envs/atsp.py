"""ATSP environment — drop-in for rrnco.envs.atsp.ATSPEnv (rrnco/envs/atsp/env.py), state updates on
the HIP kernels of csrc/rr_env.hip."""
from __future__ import annotations

import torch

from .. import _lib as L
from ..tensordict_lite import TensorDict
from .base import EnvBase


class ATSPGenerator:
    """Synthetic TMAT-class instances as LazyATSPGenerator._generate_synthetic_chunk
    (rrnco/envs/atsp/generator_lazy.py:208-237): uniform coords, uniform distances, zero diagonal,
    N passes of triangle closure."""

    def __init__(self, num_loc: int = 10, min_dist: float = 0.0, max_dist: float = 1.0, tmat_class: bool = True,
                 device="cuda", **unused):
        self.num_loc, self.min_dist, self.max_dist, self.tmat_class = num_loc, min_dist, max_dist, tmat_class
        self.device = device

    def __call__(self, batch_size, generator=None):
        bs = [batch_size] if isinstance(batch_size, int) else list(batch_size)
        n = self.num_loc
        locs = torch.rand(*bs, n, 2, device=self.device, generator=generator)
        dms = torch.rand(*bs, n, n, device=self.device, generator=generator) * (self.max_dist - self.min_dist) + self.min_dist
        ar = torch.arange(n, device=self.device)
        dms[..., ar, ar] = 0
        if self.tmat_class:
            for i in range(n):
                dms = torch.minimum(dms, dms[..., :, [i]] + dms[..., [i], :])
        return TensorDict({"locs": locs, "distance_matrix": dms}, batch_size=bs)


class ATSPEnv(EnvBase):
    name = "atsp"

    def __init__(self, generator=None, generator_params: dict = {}, normalize: bool = True, **kwargs):
        super().__init__(**kwargs)
        if generator is None:
            gp = {k: v for k, v in dict(generator_params).items() if k != "_target_"}
            generator = ATSPGenerator(**gp)
        self.generator = generator
        self.normalize = normalize

    def _reset(self, td, batch_size=None) -> TensorDict:
        """env.py:107-155."""
        distance = td["distance_matrix"]
        L.require_gpu(distance)
        dev = distance.device
        B, n = distance.shape[0], distance.shape[-1]
        out = {}
        if self.normalize:
            distance = distance.contiguous().float()
            norm = torch.empty_like(distance)
            mn = torch.empty(B, device=dev, dtype=torch.float32)
            mx = torch.empty(B, device=dev, dtype=torch.float32)
            L.check(L.lib().rr_minmax_normalize(L.ptr(distance), L.ptr(norm), L.ptr(mn), L.ptr(mx), B, n * n, L.stream()),
                    "rr_minmax_normalize")
            distance = norm
            out.update(min_distance=mn, max_distance=mx)
        cur = torch.zeros((*batch_size, 1), dtype=torch.int64, device=dev)
        out.update(distance_matrix=distance, first_node=cur, current_node=cur,
                   i=torch.zeros((*batch_size, 1), dtype=torch.int64, device=dev),
                   action_mask=torch.ones((*batch_size, n), dtype=torch.bool, device=dev))
        if td.get("locs", None) is not None:
            out["locs"] = td["locs"]
        if td.get("sample_idx", None) is not None:   # explicit neighbour-sample indices
            out["sample_idx"] = td["sample_idx"]
        note = getattr(td, "meta", {}).get("num_augment") if hasattr(td, "meta") else None      # StateAugmentation's note survives the reset
        return TensorDict(out, batch_size=batch_size, meta={"i": 0, **({"num_augment": note} if note else {})})

    def _step(self, td: TensorDict) -> TensorDict:
        """env.py:80-105; `first_node` is fixed by the first step (tracked host-side instead of the
        reference's per-step `.item()` sync)."""
        action = td["action"].contiguous()
        mask = td["action_mask"].contiguous()
        R, n = mask.shape
        steps = td.meta.get("i")
        if steps is None:
            steps = int(td["i"].flatten()[0].item())
        first = action if steps == 0 else td["first_node"]
        new_mask = torch.empty_like(mask)
        done = torch.empty(R, dtype=torch.bool, device=mask.device)
        L.check(L.lib().rr_atsp_step(L.ptr(action), L.ptr(mask), L.ptr(new_mask), L.ptr(done), R, n, L.stream()),
                "rr_atsp_step")
        td.update({"first_node": first, "current_node": action, "i": td["i"] + 1, "action_mask": new_mask,
                   "reward": torch.zeros_like(done), "done": done})
        td.meta["i"] = steps + 1
        return td

    def _get_reward(self, td: TensorDict, actions: torch.Tensor):
        """env.py:192-211 — (real, normalized) tuple when normalize."""
        D = td["distance_matrix"].contiguous()
        actions = actions.contiguous()
        R, T = actions.shape
        Bp, n = D.shape[0], D.shape[-1]
        nd = torch.empty(R, device=D.device, dtype=torch.float32)
        real = torch.empty_like(nd)
        mn = td["min_distance"].contiguous() if self.normalize else None
        mx = td["max_distance"].contiguous() if self.normalize else None
        L.check(L.lib().rr_tour_cost(L.ptr(D), L.ptr(actions), L.ptr(mn), L.ptr(mx), L.ptr(nd), L.ptr(real),
                                     R, Bp, n, T, 0, None, L.stream()), "rr_tour_cost")
        return (real, nd) if self.normalize else nd

    @staticmethod
    def check_solution_validity(td, actions):
        """env.py:213-220."""
        ar = torch.arange(actions.size(1), device=actions.device).view(1, -1).expand_as(actions)
        assert (ar == actions.sort(1)[0]).all(), "Invalid tour"
