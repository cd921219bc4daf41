"""Real-world instance sampler on the device — counterpart of rrnco/envs/{atsp,rcvrp,rmtvrp}/sampler.py.

The reference draws node subsets of a city with numpy on the host and slices the city's distance / duration matrices by
fancy indexing for every training batch (100 k instances per epoch, configs/experiment/rrnet.yaml:40).  Here the city stays
resident in HBM, the subsets are drawn with torch's device generator and the B sub-matrices are produced by one gather
kernel (csrc/rr_env.hip:k_submatrix_gather).  Same options and output keys (`points`, `distance_matrix`, and
`duration_matrix` when `with_duration`), same outlier filtering rule (entries > 1e5, sampler.py:41-60)."""
from __future__ import annotations

import torch

from .. import _lib as L


class RealWorldSampler:
    def __init__(self, with_duration: bool = False):
        self.with_duration = with_duration          # rmtvrp/sampler.py:80 also slices `duration`; atsp / rcvrp skip it
        self._city = None

    # ---- city preparation (host side, once per city)
    @staticmethod
    def filter_outliers(data: dict) -> dict:
        """sampler.py:41-60: drop the points whose rows / columns carry unreachable (> 1e5) distances."""
        dist = data["distance"]
        if float(dist.max()) <= 1e5:
            return data
        n = dist.shape[0]
        bad = dist > 1e5
        rows = torch.nonzero(bad)
        prob_r = prob_c = None
        for i in range(n):                                   # first row with a minority of bad entries -> its bad columns
            cnt = int(bad[i].sum())
            if 0 < cnt < n // 2:
                prob_r = rows[rows[:, 0] == i][:, 1]
                break
        for i in range(n):                                   # first column with a minority of bad entries -> its bad rows
            cnt = int(bad[:, i].sum())
            if cnt < n // 2:
                prob_c = rows[rows[:, 1] == i][:, 0]
                break
        drop = torch.cat([t for t in (prob_r, prob_c) if t is not None]) if (prob_r is not None or prob_c is not None) else rows[:0, 0]
        keep = torch.ones(n, dtype=torch.bool, device=dist.device)
        keep[drop] = False
        k = torch.nonzero(keep).flatten()
        out = {"points": data["points"][k], "distance": dist[k][:, k]}
        if "duration" in data:
            out["duration"] = data["duration"][k][:, k]
        return out

    def load_city(self, data: dict, device="cuda") -> None:
        """data: {"points" [M,2], "distance" [M,M], optional "duration" [M,M]} (numpy arrays or tensors)."""
        t = {k: torch.as_tensor(v) for k, v in data.items() if k in ("points", "distance", "duration")}
        t = self.filter_outliers(t)
        self._city = {k: v.to(device=device, dtype=torch.float32).contiguous() for k, v in t.items()}
        L.require_gpu(self._city["distance"])

    # ---- index sets (sampler.py:98-150)
    @staticmethod
    def uniform_indices(batch, data_length, num_sample, device, generator=None):
        """`np.random.choice(M, n, replace=False)` per instance: the first n entries of a random permutation."""
        return torch.rand(batch, data_length, device=device, generator=generator).argsort(dim=1)[:, :num_sample].contiguous()

    @staticmethod
    def single_cluster_indices(points, batch, num_sample, generator=None):
        """The num_sample points nearest to one random centre; one set shared by the batch (sampler.py:107-111)."""
        c = points[torch.randint(points.shape[0], (1,), device=points.device, generator=generator)]
        idx = (points - c).norm(dim=1).argsort()[:num_sample]
        return idx.unsqueeze(0).expand(batch, -1).contiguous()

    def mixed_indices(self, points, batch, num_sample, generator=None):
        """Per instance, num_sample of (its uniform set ++ the cluster set) without replacement (sampler.py:136-150)."""
        M = points.shape[0]
        uni = self.uniform_indices(batch, M, num_sample, points.device, generator)
        clu = self.single_cluster_indices(points, batch, num_sample, generator)
        both = torch.cat([uni, clu], dim=1)
        pick = torch.rand(batch, 2 * num_sample, device=points.device, generator=generator).argsort(dim=1)[:, :num_sample]
        return both.gather(1, pick).contiguous()

    # ---- the sampler proper
    def sample(self, batch: int, num_sample: int, loc_dist: str = "uniform", data: dict = None, generator=None, indices=None) -> dict:
        if batch <= 0 or num_sample <= 0:
            raise ValueError("batch and num_sample must be positive integers.")
        if data is not None:
            self.load_city(data)
        if self._city is None:
            raise ValueError("no city loaded: call load_city(data) or pass data=")
        pts, dist = self._city["points"], self._city["distance"]
        M = pts.shape[0]
        if num_sample > M:
            raise ValueError(f"num_sample ({num_sample}) exceeds the available data size ({M}).")
        if indices is None:
            if loc_dist == "uniform":
                indices = self.uniform_indices(batch, M, num_sample, pts.device, generator)
            elif loc_dist == "single_cluster":
                indices = self.single_cluster_indices(pts, batch, num_sample, generator)
            elif loc_dist == "mixed":
                indices = self.mixed_indices(pts, batch, num_sample, generator)
            else:                                   # "multiple_cluster" returns a ragged index set in the reference
                raise ValueError(f"Invalid loc_dist: {loc_dist}")
        indices = indices.to(device=pts.device, dtype=torch.int64).contiguous()
        out_d = torch.empty(batch, num_sample, num_sample, device=pts.device, dtype=torch.float32)
        dur = self._city.get("duration") if self.with_duration else None
        out_t = torch.empty_like(out_d) if dur is not None else None
        L.check(L.lib().rr_submatrix_gather(L.ptr(dist), L.ptr(dur), L.ptr(indices), L.ptr(out_d), L.ptr(out_t),
                                            batch, M, num_sample, L.stream()), "rr_submatrix_gather")
        res = {"points": pts[indices], "distance_matrix": out_d}
        if out_t is not None:
            res["duration_matrix"] = out_t
        return res
