"""StateAugmentation with the dihedral-8 group — drop-in for rrnco/models/utils/transforms.py:15-47,106-154
(the configuration test.py:28 builds: augment_fn='dihedral8', no_aug_coords=False, first_aug_identity=True)."""
from __future__ import annotations

import torch

from ..ops import batchify
from ..tensordict_lite import STATIC_KEYS, TensorDict


def dihedral_8_augmentation(xy: torch.Tensor) -> torch.Tensor:
    x, y = xy.split(1, dim=2)
    zs = ((x, y), (1 - x, y), (x, 1 - y), (1 - x, 1 - y), (y, x), (1 - y, x), (y, 1 - x), (1 - y, 1 - x))
    return torch.cat([torch.cat(z, dim=2) for z in zs], dim=0)


class StateAugmentation:
    def __init__(self, num_augment: int = 8, augment_fn="dihedral8", first_aug_identity: bool = True,
                 normalize: bool = False, feats=None, no_aug_coords: bool = True):
        if augment_fn != "dihedral8" or num_augment != 8 or normalize or not first_aug_identity:
            raise NotImplementedError("rrnco_amd implements the dihedral8 x8 augmentation used by test.py / rrnet.yaml")
        self.feats = [] if no_aug_coords else (["locs"] if feats is None else feats)
        self.num_augment = num_augment

    def __call__(self, td: TensorDict) -> TensorDict:
        """Augmented instances are distinct encoder inputs, so per-instance keys ARE replicated here
        (unlike multistart batchify)."""
        b = td.batch_size[0]
        out = {}
        for k, v in td.items():
            s = v.shape
            out[k] = v.expand(self.num_augment, *s).contiguous().view(s[0] * self.num_augment, *s[1:])
        for feat in self.feats:
            out[feat] = dihedral_8_augmentation(out[feat][:b])
        return TensorDict(out, batch_size=[b * self.num_augment, *td.batch_size[1:]], meta=td.meta)
