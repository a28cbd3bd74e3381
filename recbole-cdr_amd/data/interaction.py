"""Minimal stand-in for recbole's ``Interaction`` (third-party): a dict of equal-length tensors with
``[str]`` / ``[slice]`` access, ``to``, ``update`` and ``__len__`` -- the input contract of ``calculate_loss``."""


class Interaction(dict):
    def __getitem__(self, key):
        if isinstance(key, str):
            return dict.__getitem__(self, key)
        return Interaction({k: v[key] for k, v in self.items()})

    def to(self, device):
        return Interaction({k: v.to(device) for k, v in self.items()})

    def update(self, other):
        dict.update(self, other)
        return self

    def __len__(self):
        for v in self.values():
            return int(v.shape[0])
        return 0

    @property
    def length(self):
        return len(self)

    def repeat(self, times):
        return Interaction({k: v.repeat(times, *([1] * (v.dim() - 1))) for k, v in self.items()})

    def index_select(self, idx):
        return Interaction({k: v[idx] for k, v in self.items()})
