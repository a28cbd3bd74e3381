"""Load a golden fixture (tests/golden/*.npz) into the dict shapes the oracle and the product take."""
import glob
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def cases(prefix):
    return sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, prefix + '*.npz')))


class Golden:
    def __init__(self, name):
        self.name = name
        self.z = np.load(os.path.join(GOLDEN, name + '.npz'), allow_pickle=False)

    def group(self, prefix, as_torch=True, requires_grad=False):
        out = {}
        for k in self.z.files:
            if k.startswith(prefix + '/'):
                v = self.z[k]
                if as_torch:
                    v = torch.from_numpy(np.array(v))
                    if requires_grad and v.dtype.is_floating_point:
                        v.requires_grad_(True)
                out[k[len(prefix) + 1:]] = v
        return out

    def meta(self, key):
        v = self.z['meta/' + key]
        return v.item() if v.ndim == 0 else v

    def has(self, key):
        return key in self.z.files

    def __getitem__(self, key):
        return self.z[key]

    def idspace(self):
        from oracle.common import IdSpace
        return IdSpace(*(int(self.meta(k)) for k in ('OU', 'TOU', 'SOU', 'OI', 'TOI', 'SOI')))
