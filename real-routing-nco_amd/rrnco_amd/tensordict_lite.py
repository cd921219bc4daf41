"""Minimal state container with the subset of the tensordict API the RRNCO hot path uses
(SURVEY.md §8b): td[key], get/set/update, keys, batch_size/shape/size/dim, device, clone, to, td[index].

Difference from tensordict that matters for memory: keys listed in `static_keys` are per-INSTANCE data
(distance matrices, coordinates, demands ...).  `ops.batchify` does not replicate them S times — the
reference materialises `distance_matrix` as [S*B,N,N] (rrnco/models/decoding.py:189), 16 GB at the
benchmark size; here rollout r reads instance r % B inside the kernels.
"""
from __future__ import annotations

import torch

STATIC_KEYS = ("distance_matrix", "duration_matrix", "locs", "min_distance", "max_distance", "demand",
               "demand_linehaul", "demand_backhaul", "time_windows", "service_time", "sample_idx")


class TensorDict:
    def __init__(self, source=None, batch_size=None, device=None, static_repeat: int = 1, meta=None):
        if isinstance(source, TensorDict):
            static_repeat = source.static_repeat
            meta = dict(source.meta) if meta is None else meta
            source = dict(source._d)
        self._d = dict(source or {})
        if batch_size is None:
            batch_size = []
        if isinstance(batch_size, int):
            batch_size = [batch_size]
        self.batch_size = torch.Size(batch_size)
        self.static_repeat = static_repeat   # batch_size[0] == static_repeat * instances
        self.meta = dict(meta or {})         # host-side scalars (e.g. the step counter `i` without a device sync)
        if device is not None:
            self._d = {k: v.to(device) for k, v in self._d.items()}

    def is_static(self, key) -> bool:
        return self.static_repeat > 1 and key in STATIC_KEYS

    def __getitem__(self, key):
        if isinstance(key, str):
            return self._d[key]
        assert self.static_repeat == 1, "row-indexing a batchified state is not supported"
        probe = torch.empty(self.batch_size, device="meta")[key]
        return TensorDict({k: v[key] for k, v in self._d.items()}, batch_size=probe.shape)

    def __setitem__(self, key, value):
        self._d[key] = value
        if key in ("distance_matrix", "duration_matrix"):        # a replaced matrix voids StateAugmentation's "copies share the matrices" note
            self.meta.pop("num_augment", None)

    def __contains__(self, key):
        return key in self._d

    def get(self, key, default=None):
        return self._d.get(key, default)

    def set(self, key, value, inplace=False):
        self[key] = value
        return self

    def update(self, other, **kw):
        for k, v in (other._d if isinstance(other, TensorDict) else other).items():
            self[k] = v
        return self

    def index_rollouts(self, idx):
        """td[idx] for a batchified state (beam search: decoding.py:418): rows of the per-rollout keys are re-indexed, the
        per-instance keys stay as they are — valid as long as idx keeps every rollout on its own instance (r % B)."""
        if self.static_repeat == 1:
            return self[idx]
        out = {k: (v if self.is_static(k) else v[idx]) for k, v in self._d.items()}
        bs = [int(idx.shape[0]), *self.batch_size[1:]]
        return TensorDict(out, batch_size=bs, static_repeat=max(int(idx.shape[0]) // max(self.batch_size[0] // self.static_repeat, 1), 1),
                          meta=self.meta)

    def keys(self, *a, **k):
        return self._d.keys()

    def items(self):
        return self._d.items()

    def pop(self, key, default=None):
        return self._d.pop(key, default)

    @property
    def shape(self):
        return self.batch_size

    def size(self, dim=None):
        return self.batch_size if dim is None else self.batch_size[dim]

    def dim(self):
        return len(self.batch_size)

    def is_empty(self):
        return len(self._d) == 0

    @property
    def device(self):
        for v in self._d.values():
            return v.device
        return torch.device("cpu")

    def to(self, device):
        return TensorDict({k: v.to(device) for k, v in self._d.items()}, batch_size=self.batch_size,
                          static_repeat=self.static_repeat, meta=self.meta)

    def clone(self, recurse=True):
        return TensorDict({k: v.clone() for k, v in self._d.items()}, batch_size=self.batch_size,
                          static_repeat=self.static_repeat, meta=self.meta)

    def __repr__(self):
        items = ", ".join(f"{k}: {tuple(v.shape)}" for k, v in self._d.items())
        return f"TensorDict({{{items}}}, batch_size={list(self.batch_size)}, static_repeat={self.static_repeat})"
