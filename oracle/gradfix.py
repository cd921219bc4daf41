"""Compact golden vectors of a parameter gradient (TEST INFRASTRUCTURE: used by oracle/gen_golden.py and tests/ only).

A policy has ~3.4 M parameters: a full fp32 gradient is 13.5 MB per fixture, too much to commit six of.  A fixture therefore keeps,
per parameter tensor in `named_parameters()` order,
  * its Euclidean norm,
  * tensors of <= FULL_MAX elements (biases, norm scales, NAB tables): every element, fp32;
  * larger tensors: K = 32 projections <g, r_k> in float64 on Rademacher vectors r_k drawn from a generator seeded by the CRC32 of the
    tensor's name (both sides rebuild the same vectors).  For any other gradient g', mean_k (<g', r_k> - <g, r_k>)^2 is an unbiased
    estimate of |g' - g|^2 with relative standard deviation sqrt(2 / K) = 25 % per tensor (the global figure sums ~120 of them:
    a few per cent).
`deviation()` returns, per tensor, that (exact or estimated) distance and the reference norm.
"""
from __future__ import annotations

import zlib

import numpy as np
import torch

FULL_MAX = 2048
K_PROJ = 32


def _signs(name: str, n: int) -> np.ndarray:
    rng = np.random.default_rng(zlib.crc32(name.encode()) ^ 0x9E3779B9)
    return rng.integers(0, 2, size=(K_PROJ, n), dtype=np.int8) * 2 - 1


def _project(name: str, g: np.ndarray) -> np.ndarray:
    flat = g.reshape(-1).astype(np.float64)
    return _signs(name, flat.size).astype(np.float64) @ flat


def compress(names, grads: dict) -> dict:
    """grads: name -> tensor (None = no gradient: stored as zeros of the parameter's size is the caller's business)."""
    fx = {"grad_names": np.array(list(names)), "grad_norm": np.zeros(len(names), dtype=np.float64),
          "grad_numel": np.zeros(len(names), dtype=np.int64)}
    for i, n in enumerate(names):
        g = grads[n].detach().cpu().float().numpy()
        fx["grad_norm"][i] = float(np.sqrt((g.astype(np.float64) ** 2).sum()))
        fx["grad_numel"][i] = g.size
        if g.size <= FULL_MAX:
            fx[f"grad_full_{i}"] = g.reshape(-1).copy()
        else:
            fx[f"grad_proj_{i}"] = _project(n, g)
    return fx


def deviation(fx: dict, grads: dict):
    """-> list of (name, distance to the fixture's gradient, fixture norm, exact?) in the fixture's order, and the global pair."""
    rows, num, den = [], 0.0, 0.0
    names = [str(n) for n in (fx["grad_names"].tolist() if hasattr(fx["grad_names"], "tolist") else fx["grad_names"])]
    norms = np.asarray(fx["grad_norm"], dtype=np.float64)
    for i, n in enumerate(names):
        g = grads[n]
        g = (g.detach().cpu().float().numpy() if isinstance(g, torch.Tensor) else np.asarray(g, dtype=np.float32)).reshape(-1)
        kf, kp = f"grad_full_{i}", f"grad_proj_{i}"
        if kf in fx:
            ref = np.asarray(fx[kf], dtype=np.float64).reshape(-1)
            d = float(np.sqrt(((g.astype(np.float64) - ref) ** 2).sum()))
            exact = True
        else:
            ref = np.asarray(fx[kp], dtype=np.float64).reshape(-1)
            d = float(np.sqrt(((_project(n, g) - ref) ** 2).mean()))
            exact = False
        rows.append((n, d, float(norms[i]), exact))
        num += d * d
        den += float(norms[i]) ** 2
    return rows, (num ** 0.5, den ** 0.5)
