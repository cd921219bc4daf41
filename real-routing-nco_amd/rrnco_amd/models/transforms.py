"""StateAugmentation — drop-in for rrnco/models/utils/transforms.py:15-154: the dihedral-8 group (what test.py:28 and
rrnet.yaml build: augment_fn='dihedral8', no_aug_coords=False, first_aug_identity=True), the random 'symmetric' rotation /
reflection (the class default), custom callables, `normalize`, `first_aug_identity=False`.  Coordinates only: torch ops on the
device (negligible next to the encoder); the per-instance matrices are replicated."""
from __future__ import annotations

import math

import torch

from ..tensordict_lite import TensorDict


def dihedral_8_augmentation(xy: torch.Tensor) -> torch.Tensor:
    """transforms.py:15-37."""
    x, y = xy.split(1, dim=2)
    zs = ((x, y), (1 - x, y), (x, 1 - y), (1 - x, 1 - y), (y, x), (1 - y, x), (y, 1 - x), (1 - y, 1 - x))
    return torch.cat([torch.cat(z, dim=2) for z in zs], dim=0)


def dihedral_8_augmentation_wrapper(xy: torch.Tensor, reduce: bool = True, *args, **kw) -> torch.Tensor:
    """transforms.py:40-47: the input arrives already replicated x8; only its first eighth is transformed."""
    xy = xy[: xy.shape[0] // 8, ...] if reduce else xy
    return dihedral_8_augmentation(xy)


def symmetric_transform(x, y, phi, offset: float = 0.5):
    """transforms.py:50-69: rotation by phi about (offset, offset); the two axes are swapped where phi > 2 pi (a reflection)."""
    c, sn = torch.cos(phi), torch.sin(phi)
    u, v = x - offset, y - offset
    rot = torch.cat((c * u - sn * v, sn * u + c * v), dim=-1)
    return torch.where(phi > 2 * math.pi, rot.flip(-1), rot) + offset


def symmetric_augmentation(xy: torch.Tensor, num_augment: int = 8, first_augment: bool = False) -> torch.Tensor:
    """transforms.py:72-87: one random angle in [0, 4 pi) per replicated instance (same torch.rand draw as the reference);
    the first block of instances keeps phi = 0 unless `first_augment`."""
    n = xy.shape[0]
    phi = 4 * math.pi * torch.rand(n, device=xy.device)
    if not first_augment:
        phi[: n // num_augment] = 0.0
    return symmetric_transform(xy[..., 0:1], xy[..., 1:2], phi.view(n, 1, 1))


def min_max_normalize(x):
    return (x - x.min()) / (x.max() - x.min())


def get_augment_function(augment_fn):
    """transforms.py:94-103."""
    if callable(augment_fn):
        return augment_fn
    if augment_fn == "dihedral8":
        return dihedral_8_augmentation_wrapper
    if augment_fn == "symmetric":
        return symmetric_augmentation
    raise ValueError(f"Unknown augment_fn: {augment_fn}. Available options: 'symmetric', 'dihedral8' or a custom callable")


class StateAugmentation:
    def __init__(self, num_augment: int = 8, augment_fn="symmetric", first_aug_identity: bool = True,
                 normalize: bool = False, feats=None, no_aug_coords: bool = True):
        self.augmentation = get_augment_function(augment_fn)
        assert not (self.augmentation == dihedral_8_augmentation_wrapper and num_augment != 8), \
            "When using the `dihedral8` augmentation function, then num_augment must be 8"
        self.feats = [] if no_aug_coords else (["locs"] if feats is None else feats)
        self.num_augment, self.normalize, self.first_aug_identity = num_augment, normalize, first_aug_identity

    def __call__(self, td: TensorDict) -> TensorDict:
        """transforms.py:142-154.  Augmented instances are distinct encoder inputs, so per-instance keys ARE replicated here
        (unlike multistart batchify)."""
        b = td.batch_size[0]
        out = {}
        for k, v in td.items():
            s = v.shape
            out[k] = v.expand(self.num_augment, *s).contiguous().view(s[0] * self.num_augment, *s[1:])
        for feat in self.feats:
            if not self.first_aug_identity:
                init = out[feat][[b], 0].clone()          # the reference indexes with list(td.size()) = [b]: ONE element is kept
            aug = self.augmentation(out[feat], self.num_augment)
            if self.normalize:
                aug = min_max_normalize(aug)
            if not self.first_aug_identity:
                aug[[b], 0] = init
            out[feat] = aug
        # host-side note for the kernels: the batch is num_augment copies of b base instances, augmentation-major, that differ in
        # `feats` only (the matrices of copy a are those of copy 0) — the duration NAB evaluates the shared part once (rr_nab_dur_aug)
        meta = dict(td.meta)
        if self.feats in (["locs"], []):
            meta["num_augment"] = self.num_augment
        return TensorDict(out, batch_size=[b * self.num_augment, *td.batch_size[1:]], meta=meta)
