"""Dataset / checkpoint I/O around the hot path: the `.npz` test-set schema written by
scripts/generate_data.py:201-224 (rcvrp), :175-198 (rcvrptw), :354-372 (mtvrp fields), prepare_atsp_data, and the
Lightning checkpoints test.py:134-137 loads.  Pure host-side plumbing: numpy -> torch -> device."""
from __future__ import annotations

import numpy as np
import torch

from .tensordict_lite import TensorDict

# keys each problem's env.reset consumes (everything else in the file, e.g. `speed`, rides along untouched)
SCHEMA = {
    "atsp": ("locs", "distance_matrix"),
    "rcvrp": ("depot", "locs", "demand", "capacity", "distance_matrix"),
    "rcvrptw": ("locs", "distance_matrix", "duration_matrix", "demand_linehaul", "time_windows", "service_time"),
}


def load_npz_to_tensordict(path: str, device=None) -> TensorDict:
    """rl4co.data.utils.load_npz_to_tensordict (test.py:145): every array becomes a tensor, batch size = leading dim."""
    with np.load(path) as z:
        arrays = {k: z[k] for k in z.files}
    if not arrays:
        raise ValueError(f"{path}: empty archive")
    n = next(iter(arrays.values())).shape[0]
    out = {}
    for k, a in arrays.items():
        if a.ndim == 0 or a.shape[0] != n:
            raise ValueError(f"{path}: key {k!r} has shape {a.shape}, expected leading dimension {n}")
        out[k] = torch.from_numpy(np.ascontiguousarray(a))
    td = TensorDict(out, batch_size=[n])
    return td.to(device) if device is not None else td


def check_schema(td, problem: str) -> None:
    missing = [k for k in SCHEMA[problem] if k not in td]
    if missing:
        raise KeyError(f"{problem} dataset lacks {missing}; has {sorted(td.keys())}")


def prepare_for_env(td: TensorDict, problem: str) -> TensorDict:
    """The per-problem massaging test.py:152-176 does before env.reset (rcvrp: demand / capacity, capacity := 1)."""
    check_schema(td, problem)
    if problem == "rcvrp":
        td.set("demand", td["demand"] / td["capacity"].unsqueeze(-1))
        td.set("capacity", torch.ones_like(td["capacity"]))
    return td


def iter_batches(td: TensorDict, batch_size: int):
    """DataLoader(TensorDictDataset(td), batch_size, shuffle=False) of test.py:61-71: contiguous slices, last one ragged."""
    n = td.batch_size[0]
    for lo in range(0, n, batch_size):
        hi = min(n, lo + batch_size)
        yield TensorDict({k: v[lo:hi] for k, v in td.items()}, batch_size=[hi - lo])


def load_policy_state_dict(path: str) -> dict:
    """Weights of a reference checkpoint: a Lightning `.ckpt` (test.py:134-137; keys `policy.<name>` under "state_dict",
    baseline / optimizer entries dropped) or a bare policy state_dict.  Key names are the reference's."""
    blob = torch.load(path, map_location="cpu", weights_only=False)
    sd = blob.get("state_dict", blob) if isinstance(blob, dict) else blob
    if any(k.startswith("policy.") for k in sd):
        sd = {k[len("policy."):]: v for k, v in sd.items() if k.startswith("policy.")}
    return {k: v for k, v in sd.items() if torch.is_tensor(v)}


def policy_kwargs_from_state_dict(sd: dict) -> dict:
    """Architecture hyper-parameters recoverable from the tensors themselves (layers, sample size)."""
    layers = 1 + max(int(k.split(".")[3]) for k in sd if k.startswith("encoder.net.layers."))
    ss = [v.shape[1] for k, v in sd.items() if k.endswith(".row_embed.weight")][0]      # not combine_row_embed
    return dict(num_encoder_layers=layers, init_embedding_kwargs=dict(sample_size=ss))
