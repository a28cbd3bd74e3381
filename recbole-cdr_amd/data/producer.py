"""Batches produced ON the device into fixed buffers, one launch per loader per batch (csrc/cdr_sampler.hip ``batch_produce_kernel``).

What recbole's ``TrainDataLoader._next_batch_data`` + ``_neg_sampling`` do per batch for the cross-domain loaders
(recbole_cdr/data/dataloader.py:114-162; crossdomain_sampler.py:139-175) -- slice the epoch's shuffled interactions, ``repeat`` them,
draw the negatives, join -- is ~7 small launches per domain from ``DomainTrainLoader.__next__``.  A producer does the same in ONE
launch that reads its position from a device cursor and advances it itself, so the launch can be captured at the head of a training
step's hipGraph (graph_step.GraphedTrainStep): an epoch is then a run of graph replays with no per-batch host work and no batch
copy.  The host keeps the loader's own ``pr`` in step (``advance``), so ragged tails, the BOTH-mode source wrap and the end of the
epoch stay with the loader's ``__next__`` (``resync`` puts the device cursor back on ``pr`` afterwards).
"""
import torch

from .. import binding as B_
from ..utils import InputType
from .interaction import Interaction


def launch_jobs(jobs):
    arr = (B_.BatchJob * len(jobs))(*jobs)
    B_.call('cdr_batch_produce_jobs', B_.stream(), arr, len(jobs))


class DeviceBatchProducer:
    """Producer of one ``DomainTrainLoader`` (pairwise / pointwise batches with device-sampled negatives) or one ``OverlapDataloader``
    (``[OB, 1]`` id slices).  ``fields``: the Interaction of fixed tensors every launch overwrites."""

    def __init__(self, loader):
        self.loader = loader
        self.S = int(loader.step)
        if hasattr(loader, 'ids'):                                   # OverlapDataloader
            loader.pin()
            dev = loader.ids.device
            self.kind, self.k, self.pointwise = 'overlap', 0, 0
            self.users_all, self.items_all = loader.ids, None
            self.out_users = torch.zeros(self.S, 1, device=dev, dtype=torch.int64)
            self.out_items = self.out_neg = None
            self.fields = Interaction({loader.field: self.out_users})
            self.sampler = None
        else:
            loader.pin()
            smp = loader.neg_sampler
            self.users_all, self.items_all = loader.inter[loader.uid_field], loader.inter[loader.iid_field]
            dev = self.users_all.device
            self.sampler = smp
            self.k = int(loader.neg_k)
            self.pointwise = 0 if loader.input_type == InputType.PAIRWISE else 1
            self.kind = 'pointwise' if self.pointwise else 'pairwise'
            n = self.S * (1 + self.k if self.pointwise else self.k)
            self.out_users = torch.zeros(n, device=dev, dtype=torch.int64)
            self.out_items = torch.zeros(n, device=dev, dtype=torch.int64)
            if self.pointwise:
                self.out_neg = None
                lab = torch.zeros(n, device=dev, dtype=torch.float32)
                lab[:self.S] = 1.0                                    # recbole's pointwise layout: [1] * S + [0] * (S k); never changes
                self.fields = Interaction({loader.uid_field: self.out_users, loader.iid_field: self.out_items, loader.label_field: lab})
                self.fields.point_k = self.k
            else:
                self.out_neg = torch.zeros(n, device=dev, dtype=torch.int64)
                self.fields = Interaction({loader.uid_field: self.out_users, loader.iid_field: self.out_items,
                                           loader.neg_iid_field: self.out_neg})
                self.fields.k_major = self.k
        self.device = dev
        self.cursor = torch.zeros(4, device=dev, dtype=torch.int64)      # {next row, draws so far, sign-in word, spare}
        self._slots = [(self.out_users, self.out_items, self.out_neg, self.fields)]
        self.resync()

    def add_slot(self):
        """A further set of output buffers (-> its index): ``launch(slot)`` then produces the next batch THERE, so a batch can be produced
        while an earlier one is still being read (graph_step's pipelined unrolled step).  The cursor is shared: batches come in loader order
        whichever slot they land in."""
        u, i, n, f = self._slots[0]
        nf = Interaction({k: (v if (k == getattr(self.loader, 'label_field', None) and self.pointwise) else torch.zeros_like(v)) for k, v in f.items()})
        if getattr(f, 'k_major', None) is not None:
            nf.k_major = f.k_major
        if getattr(f, 'point_k', None) is not None:
            nf.point_k = f.point_k
        if self.kind == 'overlap':
            self._slots.append((nf[self.loader.field], None, None, nf))
        else:
            ld = self.loader
            self._slots.append((nf[ld.uid_field], nf[ld.iid_field], None if n is None else nf[ld.neg_iid_field], nf))
        return len(self._slots) - 1

    def fields_slot(self, slot):
        return self._slots[slot][3]

    @staticmethod
    def supports(loader):
        """A loader whose data already lives on a ROCm device and whose negatives come from ``sampler.DeviceNegSampler``."""
        from ..sampler import DeviceNegSampler
        if hasattr(loader, 'ids'):
            return loader.ids.is_cuda
        inter = getattr(loader, 'inter', None)
        if inter is None or not isinstance(getattr(loader, 'neg_sampler', None), DeviceNegSampler):
            return False
        cols = [inter[loader.uid_field], inter[loader.iid_field]]
        return all(c.is_cuda and c.dtype == torch.int64 and c.dim() == 1 for c in cols) and dict.__len__(inter) == 2 and loader.neg_k >= 1      # (Interaction.__len__ counts rows)

    def job(self, slot=0):
        """This producer's arguments as a ``cdr_batch_job`` (binding.BatchJob); the tensors behind the pointers are owned by the
        producer, the loader and the sampler, so they outlive every launch."""
        smp = self.sampler
        out_users, out_items, out_neg, _ = self._slots[slot]
        J = B_.BatchJob()
        J.users_all, J.n_rows, J.cursor, J.S = self.users_all.data_ptr(), self.users_all.numel(), self.cursor.data_ptr(), self.S
        J.k, J.pointwise, J.out_users = self.k, self.pointwise, out_users.data_ptr()
        if smp is None:
            return J
        J.items_all, J.out_items = self.items_all.data_ptr(), out_items.data_ptr()
        J.out_neg = None if out_neg is None else out_neg.data_ptr()
        J.lo0, J.hi0, J.lo1, J.hi1 = smp.ranges
        if smp.distribution == 'popularity':
            J.dist, J.keys, J.prob, J.alias, J.n_keys = 1, smp.keys.data_ptr(), smp.prob.data_ptr(), smp.alias.data_ptr(), smp.keys.numel()
        J.used_indptr, J.used_indices, J.seed, J.fail_flag = smp.indptr.data_ptr(), smp.indices.data_ptr(), smp.graph_seed(), smp.fail.data_ptr()
        return J

    def launch(self, slot=0):
        """Enqueue the production of the NEXT batch on the current stream (capturable)."""
        launch_jobs([self.job(slot)])

    def jobs(self, slot=0):
        """What ``launch`` would enqueue, for a caller that produces the batch inside another launch (trainer.DenseAdam.produce_jobs)."""
        return [self.job(slot)]

    def full_ahead(self):
        """Does the loader's next batch have all ``step`` rows (what the captured launch produces)?"""
        return self.loader.pr + self.S <= self.loader.pr_end

    def full_count(self):
        """How many full batches lie ahead of the loader's position."""
        return max(self.loader.pr_end - self.loader.pr, 0) // self.S

    def advance(self, n=1):
        """Host mirror of the cursor move ``n`` launches (replays) make."""
        self.loader.pr += n * self.S

    def resync(self):
        """Device cursor := the loader's ``pr`` (after the loader itself served a batch, wrapped or started an epoch)."""
        self.cursor[0:1].fill_(int(self.loader.pr))

    def state_tensors(self):
        return [self.cursor]


class CompositeProducer:
    """The producers of one ``CrossDomainDataloader`` state: one for SOURCE / TARGET / OVERLAP, target + source for BOTH (the batch is
    the target batch updated with the source batch, recbole_cdr/data/dataloader.py:156-161)."""

    def __init__(self, parts):
        self.parts = list(parts)
        self.fields = Interaction()
        for p in self.parts:
            self.fields.update(p.fields)
        self._slots = [self.fields]

    def add_slot(self):
        f = Interaction()
        for p in self.parts:
            f.update(p.fields_slot(p.add_slot()))
        self._slots.append(f)
        return len(self._slots) - 1

    def fields_slot(self, slot):
        return self._slots[slot]

    def launch(self, slot=0):
        launch_jobs([p.job(slot) for p in self.parts])         # ONE launch: grid row i produces loader i's batch

    def jobs(self, slot=0):
        return [p.job(slot) for p in self.parts]

    def full_ahead(self):
        return all(p.full_ahead() for p in self.parts)

    def full_count(self):
        return min(p.full_count() for p in self.parts)

    def advance(self, n=1):
        for p in self.parts:
            p.advance(n)

    def resync(self):
        for p in self.parts:
            p.resync()

    def state_tensors(self):
        return [t for p in self.parts for t in p.state_tensors()]
