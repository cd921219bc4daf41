"""rl4co.utils.ops equivalents used by callers of the policy (rl4co 0.6.0 is not vendored in the reference;
call sites: rrnco/models/decoding.py:189,203, rrnco/models/rl.py:112-121, test.py:200,211)."""
from __future__ import annotations

import torch

from .tensordict_lite import STATIC_KEYS, TensorDict


def _batchify_single(x, repeats):
    if isinstance(x, TensorDict):
        out = {}
        for k, v in x.items():
            out[k] = v if k in STATIC_KEYS else _batchify_single(v, repeats)
        return TensorDict(out, batch_size=[x.batch_size[0] * repeats, *x.batch_size[1:]],
                          static_repeat=x.static_repeat * repeats, meta=x.meta)
    s = x.shape
    return x.expand(repeats, *s).contiguous().view(s[0] * repeats, *s[1:])


def batchify(x, shape):
    """repeat-major: out[r*B + b] = x[b]; tuple shapes applied right-to-left."""
    shape = [shape] if isinstance(shape, int) else shape
    for s in reversed(shape):
        x = _batchify_single(x, s) if s > 0 else x
    return x


def _unbatchify_single(x, repeats):
    s = x.shape
    return x.view(repeats, s[0] // repeats, *s[1:]).permute(1, 0, *range(2, len(s) + 1))


def unbatchify(x, shape):
    """'(r b) ... -> b r ...'; tuple shapes applied right-to-left."""
    shape = [shape] if isinstance(shape, int) else shape
    for s in reversed(shape):
        x = _unbatchify_single(x, s) if s > 0 else x
    return x


def gather_by_index(src, idx, dim=1, squeeze=True):
    expanded_shape = list(src.shape)
    expanded_shape[dim] = -1
    idx = idx.view(idx.shape + (1,) * (src.dim() - idx.dim())).expand(expanded_shape)
    squeeze = idx.size(dim) == 1 and squeeze
    return src.gather(dim, idx).squeeze(dim) if squeeze else src.gather(dim, idx)


def get_log_likelihood(logprobs, actions=None, mask=None, return_sum=True):
    """rl4co.utils.decoding.get_log_likelihood (call site rrnco/models/policy.py:240-242)."""
    if mask is not None:
        logprobs = logprobs.masked_fill(~mask, 0)
    if logprobs.dim() == 3:
        logprobs = logprobs.gather(-1, actions.unsqueeze(-1)).squeeze(-1)
    return logprobs.sum(1) if return_sum else logprobs


def calculate_entropy(logprobs):
    """rl4co.utils.ops.calculate_entropy (call site rrnco/models/policy.py:248-249): logprobs [R, T, N] -> [R]."""
    logprobs = torch.nan_to_num(logprobs, nan=0.0)
    return -(logprobs.exp() * logprobs).sum(dim=-1).sum(dim=1)

