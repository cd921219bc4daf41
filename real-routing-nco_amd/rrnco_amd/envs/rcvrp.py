"""RCVRP environment — drop-in for rrnco.envs.rcvrp.RCVRPEnv (rrnco/envs/rcvrp/env.py)."""
from __future__ import annotations

import torch

from .. import _lib as L
from ..tensordict_lite import TensorDict
from .base import EnvBase


class RCVRPGenerator:
    """Synthetic instances: uniform depot / customers (rcvrp/generator_lazy.py:244-257), integer demands 1..9 over
    the capacity table of scripts/generate_data.py:42-57, and an explicit asymmetric matrix cdist*(1+0.2U)
    (the env's Euclidean fallback has the wrong shape: SURVEY App. D-4)."""
    CAPACITIES = {10: 20.0, 15: 25.0, 20: 30.0, 30: 33.0, 40: 37.0, 50: 40.0, 60: 43.0, 75: 45.0, 100: 50.0}

    def __init__(self, num_loc: int = 20, vehicle_capacity: float = 1.0, capacity=None, device="cuda", **unused):
        self.num_loc, self.vehicle_capacity, self.device = num_loc, vehicle_capacity, device
        self.capacity = capacity or self.CAPACITIES.get(num_loc, 50.0)

    def __call__(self, batch_size, generator=None):
        bs = [batch_size] if isinstance(batch_size, int) else list(batch_size)
        n, dev = self.num_loc, self.device
        depot = torch.rand(*bs, 2, device=dev, generator=generator)
        locs = torch.rand(*bs, n, 2, device=dev, generator=generator)
        pts = torch.cat([depot[:, None], locs], 1)
        D = torch.cdist(pts, pts) * (1 + 0.2 * torch.rand(*bs, n + 1, n + 1, device=dev, generator=generator))
        ar = torch.arange(n + 1, device=dev)
        D[:, ar, ar] = 0
        demand = torch.randint(1, 10, (*bs, n), device=dev, generator=generator).float() / self.capacity
        return TensorDict({"locs": locs, "depot": depot, "distance_matrix": D, "demand": demand}, batch_size=bs)


class RCVRPEnv(EnvBase):
    name = "rcvrp"

    def __init__(self, generator=None, generator_params: dict = {}, normalize: bool = True, **kwargs):
        super().__init__(**kwargs)
        if generator is None:
            generator = RCVRPGenerator(**{k: v for k, v in dict(generator_params).items() if k != "_target_"})
        self.generator, self.normalize = generator, normalize

    def _reset(self, td, batch_size=None) -> TensorDict:
        """env.py:124-181."""
        if "distance_matrix" not in td:
            raise ValueError("RCVRP needs an explicit [B,N+1,N+1] distance_matrix (reference fallback is mis-shaped)")
        distance = td["distance_matrix"]
        L.require_gpu(distance)
        dev, B, n1 = distance.device, distance.shape[0], distance.shape[-1]
        out = {}
        if self.normalize:
            distance = distance.contiguous().float()
            norm, mn, mx = torch.empty_like(distance), torch.empty(B, device=dev), torch.empty(B, device=dev)
            L.check(L.lib().rr_minmax_normalize(L.ptr(distance), L.ptr(norm), L.ptr(mn), L.ptr(mx), B, n1 * n1, L.stream()),
                    "rr_minmax_normalize")
            distance = norm
            out.update(min_distance=mn, max_distance=mx)
        depot = td["depot"].unsqueeze(1) if td["depot"].ndim == 2 else td["depot"]
        out.update(locs=torch.cat((depot, td["locs"]), dim=-2), distance_matrix=distance, demand=td["demand"],
                   current_node=torch.zeros(*batch_size, 1, dtype=torch.long, device=dev),
                   used_capacity=torch.zeros((*batch_size, 1), device=dev),
                   vehicle_capacity=torch.full((*batch_size, 1), float(self.generator.vehicle_capacity), device=dev),
                   visited=torch.zeros((*batch_size, n1), dtype=torch.uint8, device=dev))
        if td.get("sample_idx", None) is not None:
            out["sample_idx"] = td["sample_idx"]
        note = getattr(td, "meta", {}).get("num_augment") if hasattr(td, "meta") else None      # StateAugmentation's note survives the reset
        res = TensorDict(out, batch_size=batch_size, meta={"i": 0, **({"num_augment": note} if note else {})})
        res.set("action_mask", self.get_action_mask(res))
        return res

    @staticmethod
    def get_action_mask(td) -> torch.Tensor:
        """env.py:183-195 (torch ops; the per-step path recomputes it inside rr_rcvrp_step / the rollout kernel)."""
        dem = td["demand"]
        if td.static_repeat > 1:
            dem = dem.repeat(td.static_repeat, 1)
        exceeds = dem + td["used_capacity"] > td["vehicle_capacity"]
        mask_loc = td["visited"][..., 1:].to(exceeds.dtype) | exceeds
        mask_depot = (td["current_node"] == 0) & ((mask_loc == 0).int().sum(-1) > 0)[:, None]
        return ~torch.cat((mask_depot, mask_loc), -1)

    def _step(self, td: TensorDict) -> TensorDict:
        """env.py:90-122 on rr_rcvrp_step (state updated in place, mask recomputed in the same kernel)."""
        action = td["action"].contiguous()
        R = action.shape[0]
        dem = td["demand"].contiguous()
        used = td["used_capacity"].contiguous().clone()
        vis = td["visited"].contiguous().clone()
        vcap = td["vehicle_capacity"].contiguous()
        mask = torch.empty(R, vis.shape[-1], dtype=torch.bool, device=action.device)
        cur = torch.empty(R, 1, dtype=torch.long, device=action.device)
        done = torch.empty(R, dtype=torch.bool, device=action.device)
        L.check(L.lib().rr_rcvrp_step(L.ptr(action), L.ptr(dem), L.ptr(vcap), L.ptr(used), L.ptr(vis), L.ptr(mask),
                                      L.ptr(cur), L.ptr(done), R, dem.shape[0], dem.shape[-1], L.stream()), "rr_rcvrp_step")
        td.update({"current_node": cur, "used_capacity": used, "visited": vis, "reward": torch.zeros_like(done),
                   "done": done, "action_mask": mask})
        td.meta["i"] = td.meta.get("i", 0) + 1
        return td

    def _get_reward(self, td, actions):
        """env.py:197-219."""
        D = td["distance_matrix"].contiguous()
        actions = actions.contiguous()
        R, T = actions.shape
        nd = torch.empty(R, device=D.device, dtype=torch.float32)
        real = torch.empty_like(nd)
        mn = td["min_distance"].contiguous() if self.normalize else None
        mx = td["max_distance"].contiguous() if self.normalize else None
        L.check(L.lib().rr_tour_cost(L.ptr(D), L.ptr(actions), L.ptr(mn), L.ptr(mx), L.ptr(nd), L.ptr(real),
                                     R, D.shape[0], D.shape[-1], T, 1, None, L.stream()), "rr_tour_cost")
        return (real, nd) if self.normalize else nd

    @staticmethod
    def check_solution_validity(td, actions):
        """env.py:221-249: every customer exactly once, capacity never exceeded."""
        dem = td["demand"]
        R = actions.shape[0]
        if dem.shape[0] != R:
            dem = dem.repeat(R // dem.shape[0], 1)
        n = dem.size(1)
        sp = actions.sort(1)[0]
        ok = (torch.arange(1, n + 1, device=actions.device).view(1, -1).expand(R, n) == sp[:, -n:]).all() and (sp[:, :-n] == 0).all()
        assert ok, "Invalid tour"
        cap = td["vehicle_capacity"].reshape(-1, 1)[:1].expand(R, 1) if td["vehicle_capacity"].shape[0] != R else td["vehicle_capacity"].reshape(R, 1)
        d = torch.cat((-cap, dem), 1).gather(1, actions)
        used = torch.zeros(R, device=actions.device)
        for i in range(actions.size(1)):
            used = (used + d[:, i]).clamp_min(0)
            assert (used <= cap[:, 0] + 1e-5).all(), "Used more than capacity"
