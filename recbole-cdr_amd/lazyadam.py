"""Exact dense Adam over embedding tables, evaluated lazily per row (csrc/cdr_lazyadam.hip).

The reference's optimizer (recbole ``Trainer``: ``torch.optim.Adam(model.parameters())``) sweeps every table every step.  A row
that gets no gradient in an update evolves by a recurrence of its own state and the update number only, so ``DeferredRowAdam``
postpones those updates and replays them, in order, just before the row is read (``prepare``) or when every row is needed
(``flush``: evaluation, ``state_dict``).  The result is bit-identical to the dense sweep of ``trainer.DenseAdam`` -- same
operations in the same order on the same operands -- at O(batch) memory traffic per step.
"""
import ctypes
import os

import torch

from . import binding as B_


class DeferredRowAdam:
    """``tables``: nn.Parameters [rows, D] (same D).  ``table_list[i]``: which id list of ``prepare(id_lists)`` indexes table i
    (CoNet: [source_user, source_item, target_user, target_item] -> [0, 1, 0, 1] for id lists [user ids, item ids])."""

    def __init__(self, tables, table_list, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, capacity=1 << 16):
        self.tables = list(tables)
        self.table_list = list(table_list)
        assert len(self.tables) == len(self.table_list) and 1 <= len(self.tables) <= 4
        self.D = self.tables[0].shape[1]
        assert all(t.is_cuda and t.shape[1] == self.D and t.is_contiguous() for t in self.tables)
        self.lr, self.betas, self.eps, self.wd = float(lr), (float(betas[0]), float(betas[1])), float(eps), float(weight_decay)
        dev = self.tables[0].device
        self.exp_avg = [torch.zeros_like(t.data) for t in self.tables]
        self.exp_avg_sq = [torch.zeros_like(t.data) for t in self.tables]
        self.last = [torch.zeros(t.shape[0], device=dev, dtype=torch.int32) for t in self.tables]
        # per-update scalars (step size, sqrt(bias correction 2)) in a RING of ``capacity`` entries indexed by update number: a row
        # may lag at most capacity - 1 updates, which ``_bound_lag`` guarantees by flushing every table before the ring wraps onto
        # an entry some row still needs -- one O(table) sweep per ``capacity`` / 2 updates, no limit on the length of a run
        self.capacity = 1 << max(int(capacity) - 1, 1).bit_length()
        self.hp = torch.zeros(self.capacity, 2, device=dev, dtype=torch.float32)
        self._flushed_at = 0             # update number every row is known to have reached
        self.counters = torch.zeros(2, device=dev, dtype=torch.int64)
        self.step_count = 0              # host mirror of counters[0]
        self.dirty = False               # some row may be behind counters[0]
        self._sorted = None
        self._sorted_slots = {}
        self._bufs = {}
        self.pending = None              # (G tensor, [column offset per table], ld) stashed by the model's backward
        # software pipelining of an unrolled captured step (graph_step.GraphedTrainStep): ``early`` = the same triple, available as soon as
        # the model's FORWARD launch has produced the per-occurrence gradient rows (CoNet: conet_fb_kernel, for a unit upstream gradient);
        # ``apply_early()`` launches the update then, and the ``step()`` that follows the backward only does the bookkeeping
        self.early = None
        self._applied_early = False
        self.produce_jobs = None          # set by a pipelined captured step: the next apply launch also produces the loader's next batch (consumed there)
        self.defer_finish = False        # set around a pipelined captured step: the model's forward may leave its loss total to the backward's launches
        self._prepared = None            # id of the batch ``prepare`` last ran for (``prepare_once``)

    # ---- id sort (per list): the small rank sort up to 16,384 ids, the radix sort above --------------------------------------
    def _sort(self, id_lists, slot=0):
        """``id_lists``: per list one id tensor or a pair of tensors (the list is their concatenation, never materialised).  ``slot``: which
        set of key / position buffers receives the result (a sort can run ahead of the batches still using the other sets)."""
        dev = self.tables[0].device
        pairs = [x if isinstance(x, (tuple, list)) else (x, None) for x in id_lists]
        pairs = [(a.reshape(-1).contiguous().to(torch.int64), None if b is None or b.numel() == 0 else b.reshape(-1).contiguous().to(torch.int64))
                 for a, b in pairs]
        ns = [int(a.numel()) + (0 if b is None else int(b.numel())) for a, b in pairs]
        key = tuple(ns) + (('slot', slot),)
        if key not in self._bufs:
            tot = sum(ns)
            self._bufs[key] = (torch.empty(tot, device=dev, dtype=torch.int32), torch.empty(tot, device=dev, dtype=torch.int32),
                               torch.zeros(tot, device=dev, dtype=torch.int32))
        keys, perm, rank = self._bufs[key]
        offs, o = [], 0
        for n in ns:
            offs.append(o); o += n
        rows = max(t.shape[0] for t in self.tables)
        if max(ns) <= 16384 and len(ns) <= 4:
            m = len(ns)
            B_._alive.extend([t for p in pairs for t in p if t is not None])
            B_.call('cdr_sort_ids_small', B_.stream(), m, (ctypes.c_void_p * m)(*[a.data_ptr() for a, _ in pairs]),
                    (ctypes.c_int64 * m)(*[a.numel() for a, _ in pairs]), (ctypes.c_void_p * m)(*[None if b is None else b.data_ptr() for _, b in pairs]),
                    (ctypes.c_int64 * m)(*[0 if b is None else b.numel() for _, b in pairs]), (ctypes.c_int64 * m)(*offs), B_.raw(keys),
                    B_.raw(perm), B_.raw(rank), rows)
        else:
            if torch.cuda.is_current_stream_capturing() and max(ns) > 65536:
                # rocPRIM's Onesweep configuration (above ~200 k keys) is not hipGraph-replay-safe on this ROCm (EMCDR.fused_graph_key has
                # the story); refusing here makes the trainer's capture fail cleanly and the phase run eagerly
                raise RuntimeError('DeferredRowAdam: id lists of %d entries need the radix sort, which must not be captured in a hipGraph' % max(ns))
            for (a, b), n, of in zip(pairs, ns, offs):
                need = ctypes.c_size_t(0)
                B_._check(B_.load().cdr_sort_workspace_bytes(n, rows, ctypes.byref(need)), 'cdr_sort_workspace_bytes')
                ws = self._bufs.get(('ws', n))
                if ws is None or ws.numel() < need.value:
                    ws = torch.empty(int(need.value), device=dev, dtype=torch.uint8)
                    self._bufs[('ws', n)] = ws
                B_.call('cdr_sort_ids', B_.ctx(dev), B_.stream(), B_.i64(a), a.numel(), B_.i64(b), 0 if b is None else b.numel(), rows,
                        B_.raw(keys[of:of + n]), B_.raw(perm[of:of + n]), B_.raw(ws), ws.numel())
        return [(keys[of:of + n], perm[of:of + n], n) for n, of in zip(ns, offs)]

    def _ptrs(self, xs):
        return (ctypes.c_void_p * len(xs))(*[x.data_ptr() for x in xs])

    @torch.no_grad()
    def prepare(self, id_lists):
        """Before the forward pass: sort the batch's ids and bring their rows up to the update before the coming one."""
        if self._prepare_sort_small(id_lists):
            return
        self._sorted = self._sort(id_lists)
        self._launch_prepare()

    def _prepare_sort_small(self, id_lists, slot=0):
        """Short lists (the small rank sort's range): the replay and the sort's counting pass in ONE launch, the sort's scatter behind it
        (cdr_lazy_adam_prepare_sort_small: rows claimed through ``last``, so the replay no longer waits for the sort).  Same state and
        same sorted lists as ``_sort`` + ``_launch_prepare``; CDR_LZ_CLAIM=0 keeps those."""
        if os.environ.get('CDR_LZ_CLAIM', '1') == '0' or self.D % 2 or self.D > 512:
            return False
        dev = self.tables[0].device
        pairs = [x if isinstance(x, (tuple, list)) else (x, None) for x in id_lists]
        pairs = [(a.reshape(-1).contiguous().to(torch.int64), None if b is None or b.numel() == 0 else b.reshape(-1).contiguous().to(torch.int64))
                 for a, b in pairs]
        ns = [int(a.numel()) + (0 if b is None else int(b.numel())) for a, b in pairs]
        if not ns or min(ns) <= 0 or max(ns) > 16384 or len(ns) > 4:
            return False
        key = tuple(ns) + (('slot', slot),)
        if key not in self._bufs:
            tot = sum(ns)
            self._bufs[key] = (torch.empty(tot, device=dev, dtype=torch.int32), torch.empty(tot, device=dev, dtype=torch.int32),
                               torch.zeros(tot, device=dev, dtype=torch.int32))
        keys, perm, rank = self._bufs[key]
        offs, o = [], 0
        for n in ns:
            offs.append(o); o += n
        rows = max(t.shape[0] for t in self.tables)
        m, nT = len(ns), len(self.tables)
        if not torch.cuda.is_current_stream_capturing():
            self._bound_lag()
        B_._alive.extend([t for p in pairs for t in p if t is not None])
        keep = [[t.data for t in self.tables], self.exp_avg, self.exp_avg_sq, self.last]
        B_.call('cdr_lazy_adam_prepare_sort_small', B_.stream(), nT, self.D, self._ptrs(keep[0]), self._ptrs(keep[1]), self._ptrs(keep[2]),
                self._ptrs(keep[3]), (ctypes.c_int * nT)(*[int(j) for j in self.table_list]), m,
                (ctypes.c_void_p * m)(*[a.data_ptr() for a, _ in pairs]), (ctypes.c_int64 * m)(*[a.numel() for a, _ in pairs]),
                (ctypes.c_void_p * m)(*[None if b is None else b.data_ptr() for _, b in pairs]),
                (ctypes.c_int64 * m)(*[0 if b is None else b.numel() for _, b in pairs]), (ctypes.c_int64 * m)(*offs), B_.raw(keys),
                B_.raw(perm), B_.raw(rank), rows, self.lr, self.betas[0], self.betas[1], self.eps, self.wd, B_.raw(self.hp), self.capacity,
                B_.i64(self.counters), self.step_count + 1, (ctypes.c_int64 * nT)(*[int(t.shape[0]) for t in self.tables]), self._sweep_period())
        del keep
        self._sorted = [(keys[of:of + n], perm[of:of + n], n) for n, of in zip(ns, offs)]
        return True

    def _sweep_period(self):
        """Updates after which the replay's moving window has visited every row (cdr_lazy_adam_prepare_sort_small): 256 (128-512 measure alike at C3), less than the ring
        holds; CDR_LZ_SWEEP=0 switches the window off, any other number sets the period."""
        p = int(os.environ.get('CDR_LZ_SWEEP', '256'))
        return 0 if p <= 0 else max(2, min(p, self.capacity // 2))

    @torch.no_grad()
    def sort_ahead(self, id_lists, slot):
        """The id sort of ``prepare`` alone, for a batch that is not the next one to be read, into buffer set ``slot``;
        ``prepare_sorted(slot)`` later does the rest.  (The sort reads nothing the updates in between write.)"""
        self._sorted_slots[slot] = self._sort(id_lists, slot)

    @torch.no_grad()
    def prepare_sorted(self, slot):
        self._sorted = self._sorted_slots[slot]
        self._launch_prepare()

    def _launch_prepare(self):
        per = [self._sorted[j] for j in self.table_list]
        nT = len(self.tables)
        if not torch.cuda.is_current_stream_capturing():
            self._bound_lag()
        keep = [[t.data for t in self.tables], self.exp_avg, self.exp_avg_sq, self.last, [k for k, _, _ in per]]
        B_.call('cdr_lazy_adam_prepare', B_.stream(), nT, self.D, self._ptrs(keep[0]), self._ptrs(keep[1]), self._ptrs(keep[2]),
                self._ptrs(keep[3]), self._ptrs(keep[4]), (ctypes.c_int64 * nT)(*[n for _, _, n in per]), self.lr, self.betas[0],
                self.betas[1], self.eps, self.wd, B_.raw(self.hp), self.capacity, B_.i64(self.counters), self.step_count + 1)
        del keep

    @torch.no_grad()
    def apply_early(self):
        """The update of ``step()`` launched right behind the forward pass (``self.early`` set by the model's forward): valid only when
        the upstream gradient of the loss is exactly 1, which ``GraphedTrainStep`` guarantees (``loss.backward(ones)``)."""
        if self.early is None:
            return False
        self.pending, self.early = self.early, None
        self._launch_apply()
        self._applied_early = True
        return True

    @torch.no_grad()
    def step(self):
        """After the backward pass: the coming update for the batch's rows (``self.pending`` set by the model's backward)."""
        self.early = None
        if self._applied_early:                      # launched by apply_early(): the backward's stash is the same rows again
            self._applied_early = False
            self.pending = None
            if not torch.cuda.is_current_stream_capturing():
                self.on_replay()
            return
        if self.pending is None:
            return
        self._launch_apply()
        if not torch.cuda.is_current_stream_capturing():     # a capture only records the launches: replays do the bookkeeping
            self.on_replay()

    def _launch_apply(self):
        G, cols, ld = self.pending
        self.pending = None
        per = [self._sorted[j] for j in self.table_list]
        nT = len(self.tables)
        keep = [[t.data for t in self.tables], self.exp_avg, self.exp_avg_sq, self.last, [k for k, _, _ in per], [p for _, p, _ in per], G]
        gp = (ctypes.c_void_p * nT)(*[G.data_ptr() + 4 * c for c in cols])
        jobs, self.produce_jobs = self.produce_jobs, None
        if jobs:
            jarr = (B_.BatchJob * len(jobs))(*jobs)
            B_.call('cdr_lazy_adam_apply_produce', B_.stream(), nT, self.D, self._ptrs(keep[0]), self._ptrs(keep[1]), self._ptrs(keep[2]),
                    self._ptrs(keep[3]), self._ptrs(keep[4]), self._ptrs(keep[5]), (ctypes.c_int64 * nT)(*[n for _, _, n in per]), gp,
                    (ctypes.c_int64 * nT)(*([int(ld)] * nT)), self.lr, self.betas[0], self.betas[1], self.eps, self.wd, B_.raw(self.hp),
                    self.capacity, B_.i64(self.counters), jarr, len(jobs))
            del keep
            return
        B_.call('cdr_lazy_adam_apply', B_.stream(), nT, self.D, self._ptrs(keep[0]), self._ptrs(keep[1]), self._ptrs(keep[2]),
                self._ptrs(keep[3]), self._ptrs(keep[4]), self._ptrs(keep[5]), (ctypes.c_int64 * nT)(*[n for _, _, n in per]), gp,
                (ctypes.c_int64 * nT)(*([int(ld)] * nT)), self.lr, self.betas[0], self.betas[1], self.eps, self.wd, B_.raw(self.hp),
                self.capacity, B_.i64(self.counters))
        del keep

    def on_replay(self):
        """Host bookkeeping of one completed update (also called after a hipGraph replay of prepare + step)."""
        self.step_count += 1
        self.dirty = True
        self._bound_lag()

    def _bound_lag(self):
        """Keep every row less than ``capacity`` updates behind (the ring of per-update scalars): flush when half of it is used.
        Runs between steps -- also between hipGraph replays, whose launches only ever index the ring modulo its size."""
        if self.step_count - self._flushed_at >= self.capacity // 2:
            self.flush()

    @torch.no_grad()
    def flush(self):
        """Every row of every table up to the current update: what the dense sweep would have left in memory."""
        if not self.dirty:
            return
        for t, m, v, last in zip(self.tables, self.exp_avg, self.exp_avg_sq, self.last):
            B_.call('cdr_lazy_adam_flush', B_.stream(), self.D, B_.f32(t.data), B_.f32(m), B_.f32(v), B_.raw(last), t.shape[0], self.lr,
                    self.betas[0], self.betas[1], self.eps, self.wd, B_.raw(self.hp), self.capacity, B_.i64(self.counters))
        self.dirty = False
        self._flushed_at = self.step_count

    def state_dict(self):
        self.flush()
        return {'step': self.step_count, 'exp_avg': [m.clone() for m in self.exp_avg], 'exp_avg_sq': [v.clone() for v in self.exp_avg_sq]}

    def load_state_dict(self, sd):
        """Resume: every row is current at update ``step`` (``last`` = ``step`` everywhere), so no entry of the per-update ring from
        before ``step`` is ever read again; the coming prepares fill it from there."""
        self.step_count = int(sd['step'])
        for m, v, a, b in zip(self.exp_avg, self.exp_avg_sq, sd['exp_avg'], sd['exp_avg_sq']):
            m.copy_(a); v.copy_(b)
        for last in self.last:
            last.fill_(self.step_count)
        self.counters.fill_(self.step_count)
        self.dirty = False
        self._flushed_at = self.step_count
