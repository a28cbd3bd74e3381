"""Four-state training loader, mirroring recbole_cdr/data/dataloader.py:25-186 on device-resident interactions.

  DomainTrainLoader     recbole ``TrainDataLoader`` semantics for one domain (third-party): slices of
                        ``train_batch_size // times`` positives, ``repeat(times)`` + negatives laid out k-major
                        (pairwise: ``neg_<iid>`` column; pointwise: ``times = 1+k`` rows + ``<domain>_label``).
  OverlapDataloader     dataloader.py:25-52: slices of the shuffled ``arange(num_overlap)`` as ``[OB,1]``.
  CrossDomainDataloader dataloader.py:55-186: SOURCE / TARGET / BOTH / OVERLAP; BOTH = target batch updated with the
                        source batch, source wraps WITHOUT reshuffling, epoch length = target loader (Q11).
"""
import torch

from ..utils import CrossDomainDataLoaderState, InputType
from .interaction import Interaction


def _randperm(n, device, generator):
    """Epoch shuffle where the data lives: a host permutation of 12 M interactions costs 0.3 s -- ten times the epoch's
    kernels -- so without an explicit (host) generator it is drawn on the data's own device."""
    if generator is not None:
        return torch.randperm(n, generator=generator, device=generator.device).to(device)
    return torch.randperm(n, device=device)


class DomainTrainLoader:
    def __init__(self, inter, uid_field, iid_field, label_field, neg_prefix, train_batch_size, neg_k, input_type,
                 neg_sampler, shuffle=False, generator=None):
        self.inter = Interaction(inter)
        self.uid_field, self.iid_field, self.label_field = uid_field, iid_field, label_field
        self.neg_iid_field = neg_prefix + iid_field
        self.neg_k, self.input_type, self.neg_sampler = neg_k, input_type, neg_sampler
        self.times = neg_k if input_type == InputType.PAIRWISE else 1 + neg_k
        self.step = max(train_batch_size // self.times, 1)
        self.shuffle, self.generator = shuffle, generator
        self.pr = 0
        self._pinned = False

    @property
    def pr_end(self):
        return len(self.inter)

    def __len__(self):
        return (self.pr_end + self.step - 1) // self.step

    def pin(self):
        """Keep the interaction columns at fixed addresses from now on (an epoch shuffle then permutes them IN PLACE): what a
        captured batch producer (data/producer.py) reads must not move between hipGraph replays."""
        if not self._pinned:
            self.inter = Interaction({k: v.contiguous().clone() for k, v in self.inter.items()})     # (own copies: the caller's tensors stay as they are)
            self._pinned = True

    def __iter__(self):
        if self.shuffle:
            n = len(self.inter)
            perm = _randperm(n, next(iter(self.inter.values())).device, self.generator)
            if self._pinned:
                for v in self.inter.values():
                    v.copy_(v[perm])                 # the same permutation of the same order as index_select below, landed in place
            else:
                self.inter = self.inter.index_select(perm)
        return self

    def __next__(self):
        if self.pr >= self.pr_end:
            self.pr = 0
            # end of this loader's own epoch (standalone use, or the source side wrapping in BOTH mode): one host sync, so a user
            # without any free negative is reported even when no CrossDomainDataloader epoch follows
            if hasattr(self.neg_sampler, 'check_failures'):
                self.neg_sampler.check_failures()
            raise StopIteration()
        cur = self.inter[self.pr:self.pr + self.step]
        self.pr += self.step
        return self._neg_sampling(cur)

    def _neg_sampling(self, cur):
        S = len(cur)
        users, items = cur[self.uid_field], cur[self.iid_field]
        neg = self.neg_sampler(users, items, self.neg_k)             # [S*k], k-major (crossdomain_sampler.py:148-152)
        if self.input_type == InputType.PAIRWISE:
            out = cur.repeat(self.neg_k)
            out[self.neg_iid_field] = neg
            out.k_major = self.neg_k             # layout hint for the per-positive fused step: S positives tiled k times
            return out
        out = cur.repeat(self.times)
        out[self.iid_field] = torch.cat([items, neg])
        lab = torch.zeros(S * self.times, device=users.device, dtype=torch.float32)
        lab[:S] = 1.0
        out[self.label_field] = lab
        out.point_k = self.neg_k                 # layout hint for the per-positive pointwise step (fused.KMajorPointStep)
        return out


class OverlapDataloader:
    def __init__(self, num_overlap, overlap_batch_size, field='overlap', shuffle=False, device='cpu', generator=None):
        self.ids = torch.arange(num_overlap, device=device, dtype=torch.int64)     # PAD 0 included (dataset.py:694)
        self.step, self.field, self.shuffle, self.generator = overlap_batch_size, field, shuffle, generator
        self.pr = 0
        self._pinned = False

    def pin(self):
        """Fixed address for the id list (shuffled in place from now on): see DomainTrainLoader.pin."""
        if not self._pinned:
            self.ids = self.ids.contiguous().clone()
            self._pinned = True

    @property
    def pr_end(self):
        return self.ids.numel()

    def __len__(self):
        return (self.pr_end + self.step - 1) // self.step

    def __iter__(self):
        if self.shuffle:
            perm = _randperm(self.ids.numel(), self.ids.device, self.generator)
            if self._pinned:
                self.ids.copy_(self.ids[perm])
            else:
                self.ids = self.ids[perm]
        return self

    def __next__(self):
        if self.pr >= self.pr_end:
            self.pr = 0
            raise StopIteration()
        cur = self.ids[self.pr:self.pr + self.step]
        self.pr += self.step
        return Interaction({self.field: cur.view(-1, 1)})


class CrossDomainDataloader:
    def __init__(self, source_dataloader, target_dataloader, overlap_dataloader):
        self.source_dataloader, self.target_dataloader = source_dataloader, target_dataloader
        self.overlap_dataloader = overlap_dataloader
        self.state = CrossDomainDataLoaderState.BOTH

    def device_producer(self):
        """The current state's batches as ONE capturable launch per loader into fixed buffers (data/producer.py), or None when a
        loader of this state keeps its data on the host or samples with something other than ``sampler.DeviceNegSampler``."""
        from .producer import DeviceBatchProducer, CompositeProducer
        S = CrossDomainDataLoaderState
        loaders = {S.SOURCE: [self.source_dataloader], S.TARGET: [self.target_dataloader],
                   S.BOTH: [self.target_dataloader, self.source_dataloader], S.OVERLAP: [self.overlap_dataloader]}[self.state]
        cache = self.__dict__.setdefault('_producers', {})
        parts = []
        for dl in loaders:
            if id(dl) not in cache:
                cache[id(dl)] = DeviceBatchProducer(dl) if DeviceBatchProducer.supports(dl) else None
            if cache[id(dl)] is None:
                return None
            parts.append(cache[id(dl)])
        key = ('state', self.state)
        if key not in cache:
            cache[key] = CompositeProducer(parts)
        return cache[key]

    def check_samplers(self):
        """One host sync per sampler at the START and at the END of an epoch, in every mode (never in the middle of one): did the
        device sampler meet a user without any free candidate (sampler.DeviceNegSampler.check_failures)?"""
        for dl in (self.source_dataloader, self.target_dataloader):
            smp = getattr(dl, 'neg_sampler', None)
            if smp is not None and hasattr(smp, 'check_failures'):
                smp.check_failures()

    def __iter__(self):
        S = CrossDomainDataLoaderState
        self.check_samplers()
        if self.state == S.SOURCE:
            return self.source_dataloader.__iter__()
        if self.state == S.TARGET:
            return self.target_dataloader.__iter__()
        if self.state == S.BOTH:
            self.source_dataloader.__iter__()
            self.target_dataloader.__iter__()
            return self
        return self.overlap_dataloader.__iter__()

    def __next__(self):
        S = CrossDomainDataLoaderState
        if self.state == S.SOURCE and self.source_dataloader.pr >= self.source_dataloader.pr_end:
            self.target_dataloader.pr = 0
            self.source_dataloader.pr = 0
            self.check_samplers()                    # end of the epoch (the LAST epoch has no following __iter__ to do it)
            raise StopIteration()
        if self.state in (S.TARGET, S.BOTH) and self.target_dataloader.pr >= self.target_dataloader.pr_end:
            self.target_dataloader.pr = 0
            self.source_dataloader.pr = 0
            self.check_samplers()
            raise StopIteration()
        if self.state == S.OVERLAP and self.overlap_dataloader.pr >= self.overlap_dataloader.pr_end:
            self.overlap_dataloader.pr = 0
            raise StopIteration()
        return self._next_batch_data()

    def __len__(self):
        S = CrossDomainDataLoaderState
        return {S.SOURCE: len(self.source_dataloader), S.TARGET: len(self.target_dataloader),
                S.BOTH: len(self.target_dataloader), S.OVERLAP: len(self.overlap_dataloader)}[self.state]

    def _next_batch_data(self):
        S = CrossDomainDataLoaderState
        if self.state == S.SOURCE:
            return self.source_dataloader.__next__()
        if self.state == S.TARGET:
            return self.target_dataloader.__next__()
        if self.state == S.OVERLAP:
            return self.overlap_dataloader.__next__()
        try:
            source_data = self.source_dataloader.__next__()
        except StopIteration:
            source_data = self.source_dataloader.__next__()          # wraps, no reshuffle (dataloader.py:156-161)
        target_data = self.target_dataloader.__next__()
        target_data.update(source_data)
        return target_data

    def set_mode(self, state):
        if state not in set(CrossDomainDataLoaderState):
            raise NotImplementedError(f'Cross Domain data loader has no state named [{state}].')
        if self.source_dataloader.pr != 0 or self.target_dataloader.pr != 0:
            raise PermissionError('Cannot change dataloader\'s state within an epoch')
        self.state = state


class FullSortEvalLoader:
    """Full-sort evaluation batches in recbole's ``FullSortEvalDataLoader`` shape (third-party):
    yields ``(interaction{uid_field: users}, (history_rows, history_cols), positive_rows, positive_cols)`` with
    ``eval_batch_size // item_num`` (at least 1) users per batch, or ``users_per_batch`` when given.

    ``revoke=(overlap_item_num, target_only_item_num)`` turns it into the SOURCE-domain loader of
    recbole_cdr/data/dataloader.py:189-247: source item ids are non-contiguous, ``full_sort_predict`` concatenates the two
    ranges, so positives / history ``>= OI`` are shifted down by ``num_target_only_item`` (native ``cdr_revoke_map``)."""

    def __init__(self, uid_field, eval_pairs, history_pairs, item_num, eval_batch_size, device, revoke=None, users_per_batch=None):
        import numpy as np
        self.uid_field, self.device = uid_field, device
        # recbole sizes the batch so that the [U, N] score matrix stays within eval_batch_size entries (one user per batch for a
        # 10 M-item catalogue at the default 4,096).  The fused mask + top-k evaluation never forms that matrix, so a caller that
        # uses it (Trainer.evaluate does) may ask for a throughput-sized batch instead: 1,024 users per call score 36 times
        # more users per second than one at a time.
        self.step = int(users_per_batch) if users_per_batch else max(eval_batch_size // item_num, 1)
        # Built where it will live (recbole's history / positive matrices are host numpy; np.unique(axis=0) over 12 M
        # pairs took seconds): de-duplication and the (user, item) order are ONE device sort of integer keys per list.
        as_dev = lambda pairs: (torch.as_tensor(np.asarray(pairs, dtype=np.int64) if not torch.is_tensor(pairs) else pairs).to(device)
                                .reshape(-1, 2))
        ev_t, hi_t = as_dev(eval_pairs), as_dev(history_pairs) if len(history_pairs) else torch.zeros(0, 2, dtype=torch.int64, device=device)
        mul = int(max(int(ev_t[:, 1].max()) if ev_t.numel() else 0, int(hi_t[:, 1].max()) if hi_t.numel() else 0)) + 1
        ev_key = torch.unique(ev_t[:, 0] * mul + ev_t[:, 1])                              # sorted by (user, item), duplicates dropped
        ev_u, ev_i = ev_key // mul, ev_key % mul
        users_t = torch.unique(ev_u)
        hi_key = torch.unique(hi_t[:, 0] * mul + hi_t[:, 1]) if hi_t.numel() else hi_t.new_zeros(0)
        hi_u, hi_i = hi_key // mul, hi_key % mul
        # both pair lists are sorted by user, the batches are runs of consecutive evaluated users: every batch owns ONE
        # contiguous slice of each list (O(batch) per batch; masking all pairs per batch was quadratic in the eval set)
        if hi_u.numel():
            pos = torch.searchsorted(users_t, hi_u).clamp_(max=users_t.numel() - 1)
            keep = users_t[pos] == hi_u                                                    # history of evaluated users only
            hi_u, hi_i = hi_u[keep], hi_i[keep]
        self.users = users_t.cpu().numpy()
        self._users_t, self._ev_u, self._hi_u = users_t, ev_u, hi_u
        self._rank = torch.searchsorted(users_t, ev_u)                                     # position of each pair's user among the evaluated users
        self._hrank = torch.searchsorted(users_t, hi_u) if hi_u.numel() else hi_u
        ev_i, hi_i = ev_i.contiguous(), hi_i.contiguous()
        if revoke is not None:
            from .remap import revoke_map
            ev_i = revoke_map(ev_i, *revoke)
            hi_i = revoke_map(hi_i, *revoke) if hi_i.numel() else hi_i
        self._ev_i, self._hi_i = ev_i, hi_i
        self.rebatch(self.step)

    def rebatch(self, users_per_batch):
        """Cut the same evaluated users into batches of ``users_per_batch`` (the per-user results do not depend on the cut).  recbole sizes
        the batch for the [U, N] score matrix; an evaluation that never forms it (``Trainer.evaluate`` on ``full_sort_topk``) asks for a
        throughput-sized batch here: one user per call is ~100 x slower per user than 1,024."""
        import numpy as np
        self.step = max(int(users_per_batch), 1)
        starts = self._users_t[::self.step].contiguous()                                   # (a strided boundary tensor makes searchsorted warn on stderr)
        self._ev_ptr = np.append(torch.searchsorted(self._ev_u, starts, right=False).cpu().numpy(), self._ev_u.numel())
        self._hi_ptr = np.append(torch.searchsorted(self._hi_u, starts, right=False).cpu().numpy(), self._hi_u.numel())
        self._ev = (self._rank % self.step, self._ev_i)
        self._hi = (self._hrank % self.step, self._hi_i)
        return self

    def __len__(self):
        return (len(self.users) + self.step - 1) // self.step

    def __iter__(self):
        for b in range(len(self)):
            us = self._users_t[b * self.step:(b + 1) * self.step]
            e0, e1, h0, h1 = self._ev_ptr[b], self._ev_ptr[b + 1], self._hi_ptr[b], self._hi_ptr[b + 1]
            yield (Interaction({self.uid_field: us}), (self._hi[0][h0:h1], self._hi[1][h0:h1]),
                   self._ev[0][e0:e1], self._ev[1][e0:e1])
