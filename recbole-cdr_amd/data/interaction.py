"""Minimal stand-in for recbole's ``Interaction`` (third-party): a dict of equal-length tensors with
``[str]`` / ``[slice]`` access, ``to``, ``update`` and ``__len__`` -- the input contract of ``calculate_loss``.

``k_major`` (None or k) is a layout hint the pairwise loader sets (data/dataloader.py): the rows are S positives tiled k times with
k-major negatives (crossdomain_sampler.py:148-152).  It survives the operations that keep that layout (``to``, ``update``) and is
dropped by those that do not (row slicing, ``index_select``, ``repeat``).  ``point_k`` is the same hint for POINTWISE batches: S positives,
the user column tiled 1 + k times, items = [positives | k-major negatives] (kept by ``to`` and by an ``update`` of equal hints, dropped by everything else)."""


class Interaction(dict):
    k_major = None
    point_k = None

    def __getitem__(self, key):
        if isinstance(key, str):
            return dict.__getitem__(self, key)
        return Interaction({k: v[key] for k, v in self.items()})

    def to(self, device):
        out = Interaction({k: v.to(device) for k, v in self.items()})
        out.k_major = self.k_major
        out.point_k = self.point_k
        return out

    def update(self, other):
        was_empty = dict.__len__(self) == 0
        dict.update(self, other)
        op = getattr(other, 'point_k', None)
        self.point_k = op if was_empty else (self.point_k if op == self.point_k else None)
        ok = getattr(other, 'k_major', None)
        # BOTH-mode merge (dataloader.py:156-161): the hint survives only when both sides carry the SAME k (or self had no rows yet);
        # a hinted batch merged into rows without a hint -- or the reverse -- leaves a mixture that is not k-major
        if was_empty:
            self.k_major = ok
        elif ok != self.k_major:
            self.k_major = None
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
