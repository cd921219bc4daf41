"""RCVRPTW environment — drop-in for rrnco.envs.rmtvrp.RMTVRPEnv (rrnco/envs/rmtvrp/env.py, configs/env/rcvrptw.yaml).

The vrptw preset (linehaul demands, time windows, closed routes: BASELINE configs[3]) and instances that carry the other
multi-task features — backhauls (classes 1 / 2), open routes, distance limits — both run on the fused rollout kernel (two
instantiations; `td.meta["mtvrp_variant"]` selects) and, with `fused=False`, on the general step kernel
(rr_rmtvrp_step with MtvrpExtra) in the step-wise decode loop."""
from __future__ import annotations

import torch

from .. import _lib as L
from ..tensordict_lite import TensorDict
from .base import EnvBase


def _aug_note(td) -> dict:
    """StateAugmentation's host-side note (the batch is num_augment copies of the base instances that differ in `locs` only) survives
    the reset: the encoder's duration NAB shares the distance / duration part of its evaluation over the copies (rr_nab_dur_aug)."""
    n = getattr(td, "meta", {}).get("num_augment") if hasattr(td, "meta") else None
    return {"num_augment": n} if n else {}


def vrptw_capacity(n: int) -> float:
    return 30.0 + (n // 5 if n > 20 else 0)          # rmtvrp/generator.py:20-33


# rmtvrp/generator.py:37-58: which of the four optional features (open routes, time windows, distance limits, backhauls) a
# preset keeps; "all" / "single_feat" / "cvrp" draw ONE feature (or none) per instance, the others keep exactly the listed ones
VARIANT_PRESETS = {
    "all": "OTLB?", "single_feat": "OTLB?", "cvrp": "?", "ovrp": "O", "vrpb": "B", "vrpl": "L", "vrptw": "T", "ovrptw": "OT",
    "ovrpb": "OB", "ovrpl": "OL", "vrpbl": "LB", "vrpbtw": "TB", "vrpltw": "TL", "ovrpbl": "OLB", "ovrpbtw": "OTB",
    "ovrpltw": "OTL", "vrpbltw": "TLB", "ovrpbltw": "OTLB",
}


class RMTVRPGenerator:
    """Synthetic multi-task VRP instances: the synthetic branch of RMTVRPGenerator / LazyRMTVRPGenerator
    (rmtvrp/generator.py:183-351, 352-432, 445-513; generator_lazy.py:302-348) for every `variant_preset` of
    VARIANT_GENERATION_PRESETS (the published configuration uses "vrptw", configs/env/rcvrptw.yaml), plus explicit asymmetric
    distance / duration matrices.  Features a preset does not keep take the reference's defaults (`subsample_problems`):
    closed routes, time windows [0, inf) with zero service time, infinite distance limit, backhaul demand folded into linehaul."""

    def __init__(self, num_loc: int = 20, max_time: float = 4.6, variant_preset: str = "vrptw", device="cuda",
                 backhaul_ratio: float = 0.2, backhaul_class: int = 1, sample_backhaul_class: bool = False,
                 max_distance_limit: float = 2.8, **unused):
        if variant_preset not in VARIANT_PRESETS:
            raise NotImplementedError(f"variant_preset '{variant_preset}' (known: {sorted(VARIANT_PRESETS)})")
        assert backhaul_class in (1, 2)
        self.num_loc, self.max_time, self.device, self.variant_preset = num_loc, max_time, device, variant_preset
        self.backhaul_ratio, self.backhaul_class, self.sample_backhaul_class = backhaul_ratio, backhaul_class, sample_backhaul_class
        self.max_distance_limit = max_distance_limit

    def __call__(self, batch_size, generator=None):
        bs = [batch_size] if isinstance(batch_size, int) else list(batch_size)
        n, dev, B = self.num_loc, self.device, bs[0]
        rnd = lambda *s: torch.rand(*s, device=dev, generator=generator)  # noqa: E731
        locs = rnd(B, n + 1, 2)
        cd = torch.cdist(locs, locs)
        D, T = cd * (1 + 0.2 * rnd(B, n + 1, n + 1)), cd * (1 + 0.3 * rnd(B, n + 1, n + 1))
        ar = torch.arange(n + 1, device=dev)
        D[:, ar, ar] = 0; T[:, ar, ar] = 0
        tmn, tmx = T.amin(dim=(1, 2), keepdim=True), T.amax(dim=(1, 2), keepdim=True)
        T = (T - tmn) / (tmx - tmn)
        demand = (rnd(B, n) * 9).int().add(1).float() / vrptw_capacity(n)
        a, b, c = 0.15, 0.18, 0.2
        service = a + (b - a) * rnd(B, n)
        tw_len = b + (c - b) * rnd(B, n)
        d0 = (locs[:, 0:1] - locs[:, 1:]).norm(p=2, dim=-1)
        h_max = (self.max_time - service - tw_len) / d0 - 1
        tw_start = (1 + (h_max - 1) * rnd(B, n)) * d0
        z, full = torch.zeros(B, 1, device=dev), torch.full((B, 1), self.max_time, device=dev)
        tw = torch.stack((torch.cat((z, tw_start), -1), torch.cat((full, tw_start + tw_len), -1)), dim=-1)
        out = {"locs": locs, "distance_matrix": D, "duration_matrix": T, "demand_linehaul": demand,
               "time_windows": tw, "service_time": torch.cat((z, service), -1)}
        if self.variant_preset == "vrptw":              # the published preset: nothing else in the instance (as before)
            return TensorDict(out, batch_size=bs)
        # ---- the other features (generator.py:445-470, 577-612) and the per-instance choice of which to keep (:352-432)
        backhaul = (rnd(B, n) * 9).int().add(1).float() / vrptw_capacity(n)
        is_line = rnd(B, n) > self.backhaul_ratio
        backhaul, demand = backhaul * ~is_line, demand * is_line
        bclass = (torch.randint(1, 3, (B, 1), device=dev, generator=generator) if self.sample_backhaul_class
                  else torch.full((B, 1), self.backhaul_class, device=dev)).float()
        lower = 2 * d0.amax(dim=1) + 1e-6
        upper = torch.maximum(torch.full_like(lower, self.max_distance_limit), lower + 1e-6)
        limit = (lower + (upper - lower) * rnd(B))[:, None]
        keys = VARIANT_PRESETS[self.variant_preset]
        if "?" in keys:                                  # one feature (or plain CVRP, weight 0.5) per instance
            probs = torch.tensor([0.5 if k in keys else 0.0 for k in "OTLB"] + [0.5], device=dev)
            if self.variant_preset == "cvrp":
                probs = torch.tensor([0.0, 0.0, 0.0, 0.0, 1.0], device=dev)
            idx = torch.multinomial(probs.expand(B, 5), 1, generator=generator)[:, 0]
            keep = torch.zeros(B, 5, dtype=torch.bool, device=dev)
            keep[torch.arange(B, device=dev), idx] = True
        else:
            keep = torch.tensor([k in keys for k in "OTLB"], device=dev).expand(B, 4)
        ko, kt, kl, kb = keep[:, 0], keep[:, 1], keep[:, 2], keep[:, 3]
        open_route = ko[:, None].clone()
        tw = torch.where(kt[:, None, None], tw, torch.stack((torch.zeros_like(tw[..., 0]), torch.full_like(tw[..., 1], float("inf"))), -1))
        service_full = torch.where(kt[:, None], out["service_time"], torch.zeros_like(out["service_time"]))
        limit = torch.where(kl[:, None], limit, torch.full_like(limit, float("inf")))
        demand = torch.where(kb[:, None], demand, demand + backhaul)
        backhaul = torch.where(kb[:, None], backhaul, torch.zeros_like(backhaul))
        out.update(demand_linehaul=demand, demand_backhaul=backhaul, backhaul_class=bclass, distance_limit=limit,
                   open_route=open_route, time_windows=tw, service_time=service_full)
        return TensorDict(out, batch_size=bs)


class RMTVRPEnv(EnvBase):
    name = "rcvrptw"

    def __init__(self, generator=None, generator_params: dict = {}, select_start_nodes_fn="all", normalize: bool = True,
                 check_solution: bool = False, **kwargs):
        super().__init__(check_solution=check_solution, **kwargs)
        if generator is None:
            generator = RMTVRPGenerator(**{k: v for k, v in dict(generator_params).items() if k != "_target_"})
        if select_start_nodes_fn not in ("all", "random"):
            raise NotImplementedError("start-node selectors: 'all' (selectstartnodes.py:42-50) or 'random' (:37-40)")
        self.select_start_nodes_fn = select_start_nodes_fn
        self.generator, self.normalize = generator, normalize

    def get_num_starts(self, td):                      # selectstartnodes.py: AllSelectStartNodes -> num_loc
        return td["action_mask"].shape[-1] - 1

    def select_start_nodes(self, td, num_starts):      # selectstartnodes.py:42-50
        n = td["locs"].shape[-2] - 1
        if self.select_start_nodes_fn == "random":
            # RandomStartNodes (:37-40) draws torch.randint(0, N+1, (B, num_starts)); flattened here in the (start, instance)
            # order the batchified state has (the reference's [B, S] tensor does not fit its own td.set("action", .)), customers only
            return torch.randint(1, n + 1, (num_starts * td.shape[0],), device=td.device)
        return torch.arange(num_starts, device=td.device).repeat_interleave(td.shape[0]) % n + 1

    def _reset(self, td, batch_size=None) -> TensorDict:
        """env.py:217-341."""
        L.require_gpu(td["locs"])
        dev, B = td["locs"].device, td["locs"].shape[0]
        dl = torch.cat([torch.zeros_like(td["demand_linehaul"][..., :1]), td["demand_linehaul"]], dim=1)
        D = td["distance_matrix"] if "distance_matrix" in td else torch.cdist(td["locs"], td["locs"], p=2)
        n1 = D.shape[-1]
        out = {}
        if self.normalize:
            D = D.contiguous().float()
            norm, mn, mx = torch.empty_like(D), torch.empty(B, device=dev), torch.empty(B, device=dev)
            L.check(L.lib().rr_minmax_normalize(L.ptr(D), L.ptr(norm), L.ptr(mn), L.ptr(mx), B, n1 * n1, L.stream()),
                    "rr_minmax_normalize")
            D = norm
            out.update(min_distance=mn, max_distance=mx)
        ones = torch.ones_like(dl[..., :1])
        # env.py:225-257: optional multi-task features with their defaults
        db = td.get("demand_backhaul", None)
        given = [db is not None]
        db = torch.zeros_like(dl) if db is None else torch.cat([torch.zeros_like(dl[..., :1]), db.float()], dim=1)
        bclass = td.get("backhaul_class", None)
        given.append(bclass is not None)
        bclass = torch.full((*batch_size, 1), 1, dtype=torch.int32, device=dev) if bclass is None else bclass.to(torch.int32).reshape(-1, 1)
        limit = td.get("distance_limit", None)
        given.append(limit is not None)
        limit = torch.full_like(ones, float("inf")) if limit is None else limit.float().reshape(-1, 1)
        open_route = td.get("open_route", None)
        given.append(open_route is not None)
        open_route = torch.zeros_like(ones, dtype=torch.bool) if open_route is None else open_route.bool().reshape(-1, 1)
        # The rollout's instantiation is chosen from the instance data: ONE host read over the multi-task features that were GIVEN — none
        # when the batch carries none of them (the vrptw preset of BASELINE configs[3]: the defaults above are known here).  That read was
        # a synchronisation in every reset: it kept the host from running ahead of the device in the padded call form and made the
        # reset impossible to capture into a hipGraph.
        tests = [t for g_, t in zip(given, ((db != 0).any, lambda: (bclass != 1).any(), lambda: torch.isfinite(limit).any(), open_route.any)) if g_]
        variant = bool(torch.stack([t() for t in tests]).any()) if tests else False
        tw = td.get("time_windows", None)
        if tw is None:
            tw = torch.zeros_like(td["locs"]); tw[..., 1] = float("inf")
        out.update(
            locs=td["locs"], distance_matrix=D, duration_matrix=td["duration_matrix"] if "duration_matrix" in td else D / ones[:, None],
            demand_backhaul=db, demand_linehaul=dl, backhaul_class=bclass, distance_limit=limit,
            service_time=td.get("service_time", torch.zeros_like(dl)), open_route=open_route, time_windows=tw, speed=ones.clone(),
            vehicle_capacity=ones.clone(), capacity_original=ones.clone(),
            current_node=torch.zeros((*batch_size,), dtype=torch.long, device=dev),
            current_route_length=torch.zeros((*batch_size, 1), device=dev), current_time=torch.zeros((*batch_size, 1), device=dev),
            used_capacity_backhaul=torch.zeros((*batch_size, 1), device=dev),
            used_capacity_linehaul=torch.zeros((*batch_size, 1), device=dev),
            visited=torch.zeros((*batch_size, n1), dtype=torch.bool, device=dev))
        if td.get("sample_idx", None) is not None:
            out["sample_idx"] = td["sample_idx"]
        res = TensorDict(out, batch_size=batch_size, meta={"i": 0, "mtvrp_variant": variant, **_aug_note(td)})
        res.set("action_mask", self.get_action_mask(res))
        return res

    @staticmethod
    def get_action_mask(td) -> torch.Tensor:
        """env.py:343-428 in torch ops (reset-time only; per-step masks come from rr_rmtvrp_step / the rollout kernel)."""
        rep = td.static_repeat
        ex = (lambda v: v.repeat(rep, *([1] * (v.dim() - 1)))) if rep > 1 else (lambda v: v)
        cur = td["current_node"]
        R = cur.shape[0]
        bi = torch.arange(R, device=cur.device)
        D, T = ex(td["distance_matrix"]), ex(td["duration_matrix"])
        tw, sv, dl, db = ex(td["time_windows"]), ex(td["service_time"]), ex(td["demand_linehaul"]), ex(td["demand_backhaul"])
        opn, limit, bclass = ex(td["open_route"]), ex(td["distance_limit"]), ex(td["backhaul_class"])
        dist_ij, dist_j0 = D[bi, cur, :], D[:, :, 0]
        dur_ij, dur_j0 = T[bi, cur, :], T[:, :, 0]
        early, late = tw[..., 0], tw[..., 1]
        arrival = td["current_time"] + dur_ij
        reach = arrival < late
        back = (torch.max(arrival, early) + sv + dur_j0) * ~opn < late[..., 0:1]
        far = td["current_route_length"] + dist_ij + (dist_j0 * ~opn) > limit
        ex_l = dl + td["used_capacity_linehaul"] > td["vehicle_capacity"]
        ex_b = db + td["used_capacity_backhaul"] > td["vehicle_capacity"]
        missing = ((dl * ~td["visited"]).sum(-1) > 0)[..., None]
        carrying_b = db.gather(1, cur[:, None]) > 0
        ok1 = (missing & ~ex_l & ~carrying_b & (dl > 0)) | (~ex_b & (db > 0))
        ok2 = ~ex_l & ~ex_b & ~(dl > td["vehicle_capacity"] - td["used_capacity_backhaul"])
        ok = ((bclass == 1) & ok1) | ((bclass == 2) & ok2)
        can = reach & back & ok & ~far & ~td["visited"]
        can[:, 0] = ~((cur == 0) & (can[:, 1:].sum(-1) > 0))
        return can

    def _step(self, td: TensorDict) -> TensorDict:
        """env.py:155-215 on rr_rmtvrp_step."""
        action = td["action"].contiguous()
        R = action.shape[0]
        D, T = td["distance_matrix"].contiguous(), td["duration_matrix"].float().contiguous()
        dl, tw, sv = td["demand_linehaul"].contiguous(), td["time_windows"].contiguous(), td["service_time"].contiguous()
        vcap = td["vehicle_capacity"].reshape(-1).contiguous()
        if vcap.shape[0] != R:
            vcap = vcap.repeat(R // vcap.shape[0])
        cur = td["current_node"].reshape(-1).contiguous().clone()
        ctime, rlen = td["current_time"].contiguous().clone(), td["current_route_length"].contiguous().clone()
        used, vis = td["used_capacity_linehaul"].contiguous().clone(), td["visited"].contiguous().clone()
        mask = torch.empty(R, vis.shape[-1], dtype=torch.bool, device=action.device)
        done = torch.empty(R, dtype=torch.bool, device=action.device)
        extra, keep = None, None
        used_b = td["used_capacity_backhaul"]
        if td.meta.get("mtvrp_variant", False):
            used_b = used_b.contiguous().clone()
            keep = (td["demand_backhaul"].float().contiguous(), td["open_route"].reshape(-1).to(torch.uint8).contiguous(),
                    td["distance_limit"].float().reshape(-1).contiguous(), td["backhaul_class"].reshape(-1).to(torch.int32).contiguous())
            extra = L.MtvrpExtra()
            extra.demand_b, extra.used_b = L.ptr(keep[0]), L.ptr(used_b)
            extra.open_route, extra.dist_limit, extra.bclass = L.ptr(keep[1]), L.ptr(keep[2]), L.ptr(keep[3])
        col = td.meta.get("_to_depot")          # D[:, :, 0], T[:, :, 0] as contiguous [Bp, N] vectors, once per instance batch
        if col is None or col[0] != (D.data_ptr(), T.data_ptr()):
            col = ((D.data_ptr(), T.data_ptr()), D[:, :, 0].contiguous(), T[:, :, 0].contiguous())
            td.meta["_to_depot"] = col
        L.check(L.lib().rr_rmtvrp_step(L.ptr(action), L.ptr(D), L.ptr(T), L.ptr(col[1]), L.ptr(col[2]), L.ptr(dl), L.ptr(tw), L.ptr(sv), L.ptr(vcap),
                                       L.ptr(cur), L.ptr(ctime), L.ptr(rlen), L.ptr(used), L.ptr(vis), L.ptr(mask),
                                       L.ptr(done), R, D.shape[0], D.shape[-1], extra, L.stream()), "rr_rmtvrp_step")
        td.update({"current_node": cur, "current_route_length": rlen, "current_time": ctime, "done": done,
                   "reward": torch.zeros(R, device=action.device), "used_capacity_linehaul": used,
                   "used_capacity_backhaul": used_b, "visited": vis, "action_mask": mask})
        td.meta["i"] = td.meta.get("i", 0) + 1
        return td

    def _get_reward(self, td, actions):
        """env.py:430-455; on open routes the arcs into the depot cost nothing (:433, done inside the kernel instead of
        overwriting column 0 of the caller's matrix)."""
        D = td["distance_matrix"].contiguous()
        actions = actions.contiguous()
        R, T = actions.shape
        nd = torch.empty(R, device=D.device, dtype=torch.float32)
        real = torch.empty_like(nd)
        mn = td["min_distance"].contiguous() if self.normalize else None
        mx = td["max_distance"].contiguous() if self.normalize else None
        opn = td["open_route"].reshape(-1).to(torch.uint8).contiguous() if td.meta.get("mtvrp_variant", False) else None
        L.check(L.lib().rr_tour_cost(L.ptr(D), L.ptr(actions), L.ptr(mn), L.ptr(mx), L.ptr(nd), L.ptr(real),
                                     R, D.shape[0], D.shape[-1], T, 1, L.ptr(opn), L.stream()), "rr_tour_cost")
        return (real, nd) if self.normalize else nd

    @staticmethod
    def check_solution_validity(td, actions):
        raise NotImplementedError("unimplemented in the reference as well (rmtvrp/env.py:457-461)")
