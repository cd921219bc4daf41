"""Parameter containers of the VRP init embeddings — rrnco/models/env_embeddings/rcvrp.py:5-150 and rcvrptw.py
(identical except demand_init takes (demand, tw_start, tw_end, service_time), rcvrptw.py:44,51-56,94)."""
from __future__ import annotations

import torch
import torch.nn as nn


class _CoordinateExpert(nn.Module):      # rcvrp.py:105-124
    def __init__(self, E):
        super().__init__()
        self.init_embed_depot = nn.Linear(2, E)
        self.init_embed = nn.Linear(3, E)


class _DistanceExpert(nn.Module):        # rcvrp.py:127-137 (the *_combine_embed layers exist but are never called)
    def __init__(self, E, sample_size):
        super().__init__()
        self.row_embed, self.col_embed = nn.Linear(sample_size, E), nn.Linear(sample_size, E)
        self.row_combine_embed, self.col_combine_embed = nn.Linear(2 * E, E), nn.Linear(2 * E, E)


class _Gating(nn.Module):
    def __init__(self, E):
        super().__init__()
        self.gating_fc = nn.Sequential(nn.Linear(2 * E, 2 * E), nn.ReLU(), nn.Linear(2 * E, 1))


class RVRPInitEmbedding(nn.Module):
    demand_feats = 1

    def __init__(self, embed_dim, linear_bias=True, use_coords=True, use_polar_feats=True, use_dist=True,
                 use_matnet_init=True, sample_type="prob", sample_size=25):
        super().__init__()
        if sample_type not in ("prob", "random"):
            raise ValueError(f"sample_type {sample_type!r}: the reference knows 'prob' and 'random' (rcvrp.py:153-182)")
        if not (use_coords and use_dist and linear_bias):
            # not a gap of this engine: these branches do not run in the reference either.  use_dist=False: _embed_without_distance
            # concatenates the N customers' coordinates with the depot-padded demand of N + 1 rows (rcvrp.py:50-57 / :66: a shape error;
            # rcvrptw.py:51-68 likewise); use_coords=False with use_dist: forward calls self.coord_expert, which __init__ only builds
            # under use_coords (rcvrp.py:40-46 / :90: AttributeError).
            raise NotImplementedError("RVRPInitEmbedding: use_coords=False and use_dist=False fail inside the reference itself "
                                      "(rcvrp.py:40-46, 50-66, 90); rrnco_amd implements the branches that run: use_coords and use_dist, "
                                      "sample_type 'prob' or 'random'")
        E = embed_dim
        self.sample_size, self.sample_type = sample_size, sample_type
        self.coord_expert = _CoordinateExpert(E)
        self.gating_network_row, self.gating_network_col = _Gating(E), _Gating(E)
        self.demand_init = nn.Linear(self.demand_feats, E)
        self.combine_row_embed, self.combine_col_embed = nn.Linear(2 * E, E), nn.Linear(2 * E, E)
        self.distance_expert = _DistanceExpert(E, sample_size)

    def indices_for(self, distance, phase):
        """The neighbour index tensor of one forward (DistanceExpert._sample_indices, rcvrp.py:153-182)."""
        from .encoder import ATSPInitEmbedding, shared_random_indices
        if self.sample_type == "random":
            return shared_random_indices(phase, distance.shape[0], distance.shape[1], self.sample_size, distance.device)
        return ATSPInitEmbedding.sample_indices(distance, self.sample_size)

    def node_features(self, td):
        """[B, N+1, F]: demand with a zero for the depot (rcvrp.py:50-57)."""
        d = td["demand"]
        return torch.cat([torch.zeros_like(d[:, :1]), d], dim=1)[..., None].float()


class RVRPTWInitEmbedding(RVRPInitEmbedding):
    demand_feats = 4

    def __init__(self, embed_dim, **kw):
        super().__init__(embed_dim, **kw)
        self.init_embed = self.demand_init          # rcvrptw.py:44: the attribute layer is called `init_embed`
        del self.demand_init

    def node_features(self, td):
        """(demand_linehaul, tw_start, tw_end, service_time), all already depot-padded (rcvrptw.py:51-56)."""
        return torch.cat([td["demand_linehaul"][..., None], td["time_windows"], td["service_time"][..., None]], -1).float()


def make_vrp_init_embedding(env_name, embed_dim, **kw):
    return {"rcvrp": RVRPInitEmbedding, "rcvrptw": RVRPTWInitEmbedding}[env_name](embed_dim, **kw)
