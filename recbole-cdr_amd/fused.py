"""Fused row-wise training steps: forward + backward + optimizer for one batch without table-sized gradients
(FusedBPRStep: pairwise BPR; FusedPointStep: pointwise MSE / BCE; FusedMapStep: EMCDR's OVERLAP-phase mapping loss).

This is the large-table counterpart of ``loss.backward(); optimizer.step()`` in the reference's loop
(recbole_cdr/trainer/trainer.py:59-73 -> recbole ``Trainer._train_epoch``): dense ``[rows, D]`` gradients and a dense
Adam sweep are O(table) per step and impossible at BASELINE config C5 (72 GB of tables).  Here every step touches
only the batch's rows: see csrc/cdr_step.hip for the kernels and DESIGN.md for the (lazy-Adam) semantic note.
"""
import ctypes
import os

import torch

from . import binding as B_

OPT_SGD, OPT_ADAM = 0, 1


class RowwiseState:
    """Per-table optimizer state for the row-wise Adam (SGD needs no moments).  ``step`` counts the updates this TABLE
    has received -- like torch.optim.Adam's per-parameter ``state['step']`` -- so one state object can be shared by the
    step objects of several phases (SOURCE / TARGET BPR steps and the OVERLAP map step touch the same user tables)."""
    _step = 0                 # class-level defaults: objects built with __new__ (layout transposes in dimshard.py) start consistent
    _step_dev = None

    def __init__(self, table, opt):
        self.table = table
        self._step = 0
        self.exp_avg = torch.zeros_like(table) if opt == OPT_ADAM else None
        self.exp_avg_sq = torch.zeros_like(table) if opt == OPT_ADAM else None
        self._step_dev = None

    @property
    def step(self):
        return self._step

    @step.setter
    def step(self, value):                       # (checkpoint restore, layout changes) keeps the device mirror in step
        self._step = int(value)
        if self._step_dev is not None:
            self._step_dev.fill_(self._step)

    @property
    def step_dev(self):
        """The update count as a device int64 [1] (read by the capturable applies); created on first use from ``step``."""
        if self._step_dev is None:
            self._step_dev = torch.full((1,), int(self._step), device=self.table.device, dtype=torch.int64)
        return self._step_dev

    def advance(self, device_bumped=False):
        """One more update of this table.  ``device_bumped``: a kernel already incremented the device counter."""
        self._step += 1
        if self._step_dev is not None and not device_bumped:
            B_.call('cdr_inc_i64', B_.stream(), B_.i64(self._step_dev))


class FusedBPRStep:
    """One object per (user table, item table) pair; buffers are sized for ``max_batch`` triples and reused."""

    def __init__(self, user_table, item_table, max_batch, opt='adam', lr=1e-3, betas=(0.9, 0.999), eps=1e-8,
                 weight_decay=0.0, gamma=1e-10, reg_weight=0.0, user_state=None, item_state=None, fuse_singles=True, device_counts=True,
                 id_path='auto'):
        assert user_table.is_cuda and item_table.is_cuda, 'FusedBPRStep needs ROCm device tensors'
        assert user_table.shape[1] == item_table.shape[1]
        self.U, self.I = user_table, item_table
        self.D = user_table.shape[1]
        self.opt = OPT_ADAM if opt == 'adam' else OPT_SGD
        self.lr, self.betas, self.eps, self.wd = lr, betas, eps, weight_decay
        self.gamma, self.reg_weight = gamma, reg_weight
        self.ustate = user_state if user_state is not None else RowwiseState(user_table, self.opt)
        self.istate = item_state if item_state is not None else RowwiseState(item_table, self.opt)
        dev = user_table.device
        Bm = int(max_batch)
        self.max_batch = Bm
        self.GU = torch.empty(Bm, self.D, device=dev, dtype=torch.float32)
        self.GP = torch.empty(Bm, self.D, device=dev, dtype=torch.float32)
        self.out6 = torch.zeros(12, device=dev, dtype=torch.float32)
        # one sort per step for BOTH tables (the radix sort costs the same ~0.16 ms for 1 M or 3 M pairs): the user keys
        # come out first, the item keys behind them with a table bit (key_base) the apply kernel subtracts again
        self.keys = torch.empty(3 * Bm, device=dev, dtype=torch.int32)
        self.perm = torch.empty(3 * Bm, device=dev, dtype=torch.int32)
        rows = max(user_table.shape[0], item_table.shape[0])
        self._sort_rows = 2 << (rows - 1).bit_length()
        need = ctypes.c_size_t(0)
        B_._check(B_.load().cdr_sort_workspace_bytes(3 * Bm, self._sort_rows, ctypes.byref(need)), 'cdr_sort_workspace_bytes')
        self.ws_bytes = int(need.value)
        self.ws = torch.empty(self.ws_bytes, device=dev, dtype=torch.uint8)
        self._key_base = ctypes.c_uint32(0)
        # single-occurrence rows updated by the forward kernel (cdr_bpr_step_fused): D <= 256 like cdr_bpr_fwd_grad
        self.fuse_singles = bool(fuse_singles) and self.D % 4 == 0 and self.D <= 256 and os.environ.get('CDR_FUSE_SINGLES', '1') != '0'   # env: A/B runs
        # Adam update counts on the device (cdr_bpr_step_fused_dev): the step can then be captured in a hipGraph (``replayed`` keeps the
        # host mirrors in step); the host-count form stays available (device_counts=False: the sharded layouts drive it)
        self.device_counts = bool(device_counts) and self.fuse_singles
        self._hp_dev = None
        if self.fuse_singles:
            words = ctypes.c_int64(0)
            B_._check(B_.load().cdr_bpr_step_fused_heads_words(Bm, ctypes.byref(words)), 'cdr_bpr_step_fused_heads_words')
            self.flags = torch.zeros(4 * Bm, device=dev, dtype=torch.uint8)      # {user, positive, negative, -} per triple
            self.heads = torch.empty(int(words.value), device=dev, dtype=torch.int32)
        # Ids without a sort (round 6, csrc/cdr_step.hip): batches of 16,448 ... 131,072 triples take their single-occurrence flags from one
        # counter per table row and sort only the duplicate occurrences.  'auto': on, and moved to the sorted path (and back) by the
        # duplicate statistics the step leaves in heads[2:4] -- read back asynchronously, one step late, never waited for; 'count' / 'sort' pin it.
        self.id_path = os.environ.get('CDR_ID_PATH', id_path)
        assert self.id_path in ('auto', 'count', 'sort')
        self._count = None
        self._use_count = self.id_path != 'sort'
        self._stat = None

    COUNT_MIN_B, COUNT_MAX_B = 16448, 131072

    def _count_buffers(self):
        if self._count is None:
            dev = self.U.device
            need = ctypes.c_size_t(0)
            B_._check(B_.load().cdr_id_count_workspace_bytes(min(self.max_batch, self.COUNT_MAX_B), ctypes.byref(need)), 'cdr_id_count_workspace_bytes')
            self._count = (torch.zeros(self.U.shape[0], device=dev, dtype=torch.int32), torch.zeros(self.I.shape[0], device=dev, dtype=torch.int32),
                           torch.empty(int(need.value), device=dev, dtype=torch.uint8))
        return self._count

    def _select_id_path(self, B):
        """Hands the counters to (or takes them from) the native context of the current stream before a fused step; folds in the statistics
        of an EARLIER step if their copy has landed (never waits)."""
        ctxh = B_.ctx(self.U.device)
        in_range = self.fuse_singles and self.COUNT_MIN_B <= B <= self.COUNT_MAX_B
        capturing = torch.cuda.is_current_stream_capturing()
        if in_range and self.id_path == 'auto' and not capturing:
            if self._stat is not None and self._stat[1].query():
                nd, maxc = int(self._stat[0][2]), int(self._stat[0][3])
                nB = self._stat[2]
                if self._use_count and (nd > 10240 or maxc > 256):
                    self._use_count = False                  # a skewed stream: same-address counter updates and a long duplicate list -- the sort serves it better
                elif not self._use_count and nd <= 8192:     # (the sorted path reports the duplicate occurrences only: heads[3] stays 0 there)
                    self._use_count = True
                self._stat = None
        on = in_range and self._use_count and not (capturing and self._count is None)     # (never allocate + zero-fill 4 B per table row inside a capture)
        if on:
            cu, ci, ws = self._count_buffers()
            B_.call('cdr_ctx_set_id_counters', ctxh, B_.raw(cu), cu.numel(), B_.raw(ci), ci.numel(), B_.raw(ws), ws.numel())
        else:
            B_.call('cdr_ctx_set_id_counters', ctxh, None, 0, None, 0, None, 0)
        return in_range and self.id_path == 'auto' and not capturing

    def _note_stats(self, B):
        if self._stat is None:
            host = torch.empty(4, dtype=torch.int32, pin_memory=True)
            host.copy_(self.heads[:4], non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
            self._stat = (host, ev, B)

    def step(self, uid, pid, nid):
        """uid/pid/nid: int64 device tensors [B].  Returns the device tensor out6 (view; [0] = total loss)."""
        B = uid.numel()
        assert B <= self.max_batch
        if self.fuse_singles:
            return self._step_fused(uid, pid, nid, B)
        B_.call('cdr_bpr_fwd_grad', B_.ctx(self.U.device), B_.stream(), B_.f32(self.U), B_.f32(self.I), self.D, B_.i64(uid),
                B_.i64(pid), B_.i64(nid), B, 0, float(self.gamma), float(self.reg_weight), B_.f32(self.out6), B_.f32(self.GU),
                B_.f32(self.GP), 0)
        return self.sort_apply(uid, pid, nid)

    def _step_fused(self, uid, pid, nid, B):
        """Batch norms and sort first, then ONE pass that also applies the optimizer to every row occurring once in the batch; the
        segmented applies see the duplicate rows only (csrc/cdr_step.hip, "single-occurrence rows in the forward")."""
        us, its = self.ustate, self.istate
        watch = self._select_id_path(B)
        try:
            return self._step_fused_call(uid, pid, nid, B, us, its)
        finally:
            # (the context is shared by every step object of this stream: it must not keep pointers into this object's buffers)
            B_.call('cdr_ctx_set_id_counters', B_.ctx(self.U.device), None, 0, None, 0, None, 0)
            self._steps_seen = self.__dict__.get('_steps_seen', 0) + 1
            if watch and (self._steps_seen & 15) == 1:          # every 16th step: one 16-byte copy, read when it has landed
                self._note_stats(B)

    def _step_fused_call(self, uid, pid, nid, B, us, its):
        if self.opt == OPT_ADAM and self.device_counts:
            # the capturable form: the update counts live on the device and the call advances them itself (the host mirrors follow)
            if self._hp_dev is None:
                self._hp_dev = torch.zeros(4, device=self.U.device, dtype=torch.float32)
            su, si = us.step_dev, its.step_dev
            B_.call('cdr_bpr_step_fused_dev', B_.ctx(self.U.device), B_.stream(), self.opt, B_.f32(us.table), B_.f32(us.exp_avg),
                    B_.f32(us.exp_avg_sq), us.table.shape[0], B_.f32(its.table), B_.f32(its.exp_avg), B_.f32(its.exp_avg_sq),
                    its.table.shape[0], self.D, B_.i64(uid), B_.i64(pid), B_.i64(nid), B, float(self.gamma),
                    float(self.reg_weight), float(self.lr), float(self.betas[0]), float(self.betas[1]), float(self.eps), float(self.wd),
                    B_.i64(su), B_.i64(si), B_.f32(self._hp_dev), B_.f32(self.out6), B_.f32(self.GU), B_.f32(self.GP), B_.raw(self.keys),
                    B_.raw(self.perm), B_.raw(self.flags), B_.raw(self.heads), B_.raw(self.ws), self.ws_bytes)
            if not torch.cuda.is_current_stream_capturing():
                us.advance(device_bumped=True)
                its.advance(device_bumped=True)
            return self.out6
        us.advance()
        its.advance()
        B_.call('cdr_bpr_step_fused', B_.ctx(self.U.device), B_.stream(), self.opt, B_.f32(us.table), B_.f32(us.exp_avg),
                B_.f32(us.exp_avg_sq), us.table.shape[0], B_.f32(its.table), B_.f32(its.exp_avg), B_.f32(its.exp_avg_sq),
                its.table.shape[0], self.D, B_.i64(uid), B_.i64(pid), B_.i64(nid), B, float(self.gamma),
                float(self.reg_weight), float(self.lr), float(self.betas[0]), float(self.betas[1]), float(self.eps), float(self.wd),
                us.step, its.step, B_.f32(self.out6), B_.f32(self.GU), B_.f32(self.GP), B_.raw(self.keys), B_.raw(self.perm),
                B_.raw(self.flags), B_.raw(self.heads), B_.raw(self.ws), self.ws_bytes)
        return self.out6

    def replayed(self, n=1):
        """Host bookkeeping of ``n`` hipGraph replays of ``step`` (device_counts form): the update counts' host mirrors."""
        for _ in range(n):
            self.ustate.advance(device_bumped=True)
            self.istate.advance(device_bumped=True)

    def sort_apply(self, uid, pid, nid):
        """Second half of the step: GU / GP / out6[4:6] are in place (written by the forward kernel); one sort for both tables,
        two row-wise applies."""
        self.sort_ids(uid, pid, nid)
        return self.apply_sorted(uid.numel())

    def sort_ids(self, uid, pid, nid):
        """The id sort alone -- it needs nothing but the ids, so dimshard.py runs it under the all-reduce of the partial scores."""
        B = uid.numel()
        B_.call('cdr_sort_ids_two_tables', B_.ctx(self.U.device), B_.stream(), B_.i64(uid), B, self.U.shape[0], B_.i64(pid), B,
                B_.i64(nid), B, self.I.shape[0], B_.raw(self.keys), B_.raw(self.perm), ctypes.byref(self._key_base), B_.raw(self.ws),
                self.ws_bytes)

    def apply_sorted(self, B):
        ctxh = B_.ctx(self.U.device)
        self._apply(ctxh, self.ustate, self.keys[:B], self.perm[:B], B, self.GU, B, B, self.out6[4:5], 0)
        self._apply(ctxh, self.istate, self.keys[B:3 * B], self.perm[B:3 * B], 2 * B, self.GP, B, B, self.out6[5:6],
                    self._key_base.value)
        return self.out6

    def _apply(self, ctxh, st, keys, perm, n, G, neg_start, reg_limit, coef, key_base):
        st.advance()
        B_.call('cdr_rowwise_apply', ctxh, B_.stream(), self.opt, B_.f32(st.table), B_.f32(st.exp_avg),
                B_.f32(st.exp_avg_sq), self.D, B_.raw(keys), B_.raw(perm), n, B_.f32(G), neg_start, reg_limit,
                B_.f32(coef), float(self.lr), float(self.betas[0]), float(self.betas[1]), float(self.eps),
                float(self.wd), st.step, None, int(key_base))


class KMajorBPRStep:
    """The fused BPR step cut along recbole's pairwise batch layout (crossdomain_sampler.py:148-152; emcdr.py:123-131): S positives
    tiled k times with k-major negatives.  One lane group per POSITIVE in the forward (u, p gathered once, not k times), one
    gradient row per positive for the user table, and no item gradient rows at all: the item apply rebuilds each occurrence's
    row as coefficient * user row (csrc/cdr_kstep.hip).  Same loss and same per-row gradients as FusedBPRStep on the B = S k
    rows.  Batches whose item list fits the small rank sort (S k + S <= 8192) run as 4 launches behind one native call with device-side Adam
    counters and can be replayed as a hipGraph (``capture``)."""

    def __init__(self, user_table, item_table, max_positives, k=1, opt='adam', lr=1e-3, betas=(0.9, 0.999), eps=1e-8,
                 weight_decay=0.0, gamma=1e-10, reg_weight=0.0, user_state=None, item_state=None, fuse_singles=True):
        assert user_table.is_cuda and item_table.is_cuda, 'KMajorBPRStep needs ROCm device tensors'
        assert user_table.shape[1] == item_table.shape[1]
        self.U, self.I = user_table, item_table
        self.D = user_table.shape[1]
        self.k = int(k)
        self.opt = OPT_ADAM if opt == 'adam' else OPT_SGD
        self.lr, self.betas, self.eps, self.wd = lr, betas, eps, weight_decay
        self.gamma, self.reg_weight = gamma, reg_weight
        self.ustate = user_state if user_state is not None else RowwiseState(user_table, self.opt)
        self.istate = item_state if item_state is not None else RowwiseState(item_table, self.opt)
        dev = user_table.device
        Sm = int(max_positives)
        Bm = Sm * self.k
        self.max_positives = Sm
        self.GU = torch.empty(Sm, self.D, device=dev, dtype=torch.float32)
        # large batches: rows that occur once are updated by the forward kernel (cdr_bpr_step_fused_kmajor); the duplicate item
        # occurrences get their gradient row written (the user row it is made of is updated by that same kernel) instead of a record
        self.fuse_singles = (bool(fuse_singles) and (Sm + Bm) > 8192 and self.D % 4 == 0 and self.D <= 256 and self.k <= 64
                             and os.environ.get('CDR_FUSE_SINGLES', '1') != '0')
        if self.fuse_singles:
            fb, hw = ctypes.c_int64(0), ctypes.c_int64(0)
            B_._check(B_.load().cdr_bpr_step_fused_kmajor_sizes(Sm, self.k, ctypes.byref(fb), ctypes.byref(hw)), 'cdr_bpr_step_fused_kmajor_sizes')
            self.GI = torch.empty(Sm + Bm, self.D, device=dev, dtype=torch.float32)
            self.flags = torch.zeros(int(fb.value), device=dev, dtype=torch.uint8)
            self.heads = torch.empty(int(hw.value), device=dev, dtype=torch.int32)
            self.rec = None
        else:
            self.rec = torch.empty(2 * (Sm + Bm), device=dev, dtype=torch.int32)      # 8-byte {user row, coefficient} records
        self.out6 = torch.zeros(12, device=dev, dtype=torch.float32)
        n_all = 2 * Sm + Bm                                                           # user list [S] ++ item list [S + B]
        self.keys = torch.empty(n_all, device=dev, dtype=torch.int32)
        self.perm = torch.empty(n_all, device=dev, dtype=torch.int32)
        self.small = (Sm + Bm) <= 8192                # the rank sort is quadratic: past this the radix sort wins
        self.rank = torch.zeros(n_all, device=dev, dtype=torch.int32) if self.small else None   # scratch of the rank sort
        self.ws = None
        self._key_base = ctypes.c_uint32(0)
        if not self.small:
            rows = max(user_table.shape[0], item_table.shape[0])
            need = ctypes.c_size_t(0)
            B_._check(B_.load().cdr_sort_workspace_bytes(n_all, 2 << (rows - 1).bit_length(), ctypes.byref(need)),
                      'cdr_sort_workspace_bytes')
            self.ws = torch.empty(int(need.value), device=dev, dtype=torch.uint8)
        self._graph = None

    def step(self, uid, pid, nid):
        """uid / pid: int64 [S] (or the reference's tiled [S k]: the first S entries are read), nid: int64 [S k] k-major.
        Returns out6 (view; [0] = total loss)."""
        B = nid.numel()
        S = B // self.k
        assert S * self.k == B and S <= self.max_positives and uid.numel() >= S and pid.numel() >= S
        if self.fuse_singles:
            us, its = self.ustate, self.istate
            us.advance(); its.advance()
            B_.call('cdr_bpr_step_fused_kmajor', B_.ctx(self.U.device), B_.stream(), self.opt, B_.f32(us.table), B_.f32(us.exp_avg),
                    B_.f32(us.exp_avg_sq), us.table.shape[0], B_.f32(its.table), B_.f32(its.exp_avg), B_.f32(its.exp_avg_sq),
                    its.table.shape[0], self.D, B_.i64(uid), B_.i64(pid), B_.i64(nid), S, self.k, float(self.gamma), float(self.reg_weight),
                    float(self.lr), float(self.betas[0]), float(self.betas[1]), float(self.eps), float(self.wd), us.step, its.step,
                    B_.f32(self.out6), B_.f32(self.GU), B_.f32(self.GI), B_.raw(self.keys), B_.raw(self.perm), B_.raw(self.flags),
                    B_.raw(self.heads), B_.raw(self.ws), self.ws.numel())
            return self.out6
        self._enqueue(uid, pid, nid, S)
        self.ustate.advance(device_bumped=True)
        self.istate.advance(device_bumped=True)
        return self.out6

    def _enqueue(self, uid, pid, nid, S):
        dev = self.U.device
        ctxh, s = B_.ctx(dev), B_.stream()
        B = S * self.k
        adam = self.opt == OPT_ADAM
        ud, idv = (self.ustate.step_dev, self.istate.step_dev) if adam else (None, None)
        hp = (float(self.lr), float(self.betas[0]), float(self.betas[1]), float(self.eps), float(self.wd))
        if self.small:
            # four launches behind one call: {forward || rank count}, {scatter || loss finish}, item apply, user apply
            us, its = self.ustate, self.istate
            B_.call('cdr_bpr_step_small', ctxh, s, self.opt, B_.f32(us.table), B_.f32(us.exp_avg), B_.f32(us.exp_avg_sq),
                    B_.f32(its.table), B_.f32(its.exp_avg), B_.f32(its.exp_avg_sq), self.D, B_.i64(uid), B_.i64(pid), B_.i64(nid), S,
                    self.k, float(self.gamma), float(self.reg_weight), *hp, B_.i64(ud), B_.i64(idv), B_.f32(self.out6),
                    B_.f32(self.GU), B_.raw(self.rec), B_.raw(self.keys), B_.raw(self.perm), B_.raw(self.rank),
                    max(self.U.shape[0], self.I.shape[0]))
            return
        B_.call('cdr_bpr_fwd_grad_kmajor', ctxh, s, B_.f32(self.U), B_.f32(self.I), self.D, B_.i64(uid), B_.i64(pid), B_.i64(nid), S,
                self.k, float(self.gamma), float(self.reg_weight), B_.f32(self.out6), B_.f32(self.GU), B_.raw(self.rec),
                B_.i64(ud), B_.i64(idv))
        base = self._sort(uid, pid, nid, S, ctxh)
        # the item apply reads the PRE-step user rows: it runs first
        st = self.istate
        tab, m, v = self._offset(st, base)
        B_.call('cdr_rowwise_apply_scaled', ctxh, s, self.opt, tab, m, v, self.D, B_.raw(self.keys[S:2 * S + B]),
                B_.raw(self.perm[S:2 * S + B]), S + B, B_.raw(self.rec), B_.f32(self.U), S, B_.f32(self.out6[5:6]), *hp,
                st.step + 1, B_.i64(idv), 0)
        st = self.ustate
        B_.call('cdr_rowwise_apply_rows', ctxh, s, self.opt, B_.f32(st.table), B_.f32(st.exp_avg), B_.f32(st.exp_avg_sq), self.D,
                B_.raw(self.keys[:S]), B_.raw(self.perm[:S]), S, B_.f32(self.GU), S, B_.f32(self.out6[4:5]), *hp, st.step + 1,
                B_.i64(ud), 0)

    def _sort(self, uid, pid, nid, S, ctxh):
        """User list [S] at keys[0:S], item list [pid | nid] at keys[S:2S+B]; returns the item keys' table bit (0 for the rank sort)."""
        B = S * self.k
        if self.small:
            arr = lambda xs: (ctypes.c_void_p * 2)(*xs)
            i64s = lambda xs: (ctypes.c_int64 * 2)(*xs)
            B_._alive.extend([uid, pid, nid])
            B_.call('cdr_sort_ids_small', B_.stream(), 2, arr([uid.data_ptr(), pid.data_ptr()]), i64s([S, S]), arr([None, nid.data_ptr()]),
                    i64s([0, B]), i64s([0, S]), B_.raw(self.keys), B_.raw(self.perm), B_.raw(self.rank), max(self.U.shape[0], self.I.shape[0]))
            return 0
        B_.call('cdr_sort_ids_two_tables', ctxh, B_.stream(), B_.i64(uid), S, self.U.shape[0], B_.i64(pid), S, B_.i64(nid), B,
                self.I.shape[0], B_.raw(self.keys), B_.raw(self.perm), ctypes.byref(self._key_base), B_.raw(self.ws), self.ws.numel())
        return int(self._key_base.value)

    def _offset(self, st, key_base):
        """Table pointers moved back by key_base rows (keys of a two-table sort carry the table bit; see cdr_rowwise_apply)."""
        off = 4 * key_base * self.D
        f = lambda t: None if t is None else ctypes.c_void_p(t.data_ptr() - off)
        B_._alive.extend([st.table, st.exp_avg, st.exp_avg_sq])
        return f(st.table), f(st.exp_avg), f(st.exp_avg_sq)

    # ---- hipGraph replay of the whole step (small batches: the reference's default train_batch_size is 2,048) ----------
    def capture(self, S):
        """Capture the 6-launch step for batches of exactly S positives.  ``replay(uid, pid, nid)`` then costs three small id
        copies + one graph launch on the host."""
        assert self.small and self.opt in (OPT_ADAM, OPT_SGD)
        dev = self.U.device
        B = S * self.k
        self._S = S
        self._ids = (torch.zeros(S, device=dev, dtype=torch.int64), torch.zeros(S, device=dev, dtype=torch.int64),
                     torch.zeros(B, device=dev, dtype=torch.int64))
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            B_.ctx(dev)                                   # the native context of the capture stream must exist beforehand
            if self.opt == OPT_ADAM:
                self.ustate.step_dev, self.istate.step_dev
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self._graph = torch.cuda.CUDAGraph()
        with B_.capturing(self._graph, side):
            self._enqueue(self._ids[0], self._ids[1], self._ids[2], S)
        return self

    @property
    def static_ids(self):
        """(uid [S], pid [S], nid [S k]): the buffers the captured graph reads.  A producer that writes its batch straight into
        them (the device sampler / loader) can call ``replay()`` without arguments: no id copies at all."""
        return self._ids

    def replay(self, uid=None, pid=None, nid=None):
        S = self._S
        if uid is not None:
            self._ids[0].copy_(uid[:S]); self._ids[1].copy_(pid[:S]); self._ids[2].copy_(nid)
        self._graph.replay()
        self.ustate.advance(device_bumped=True)
        self.istate.advance(device_bumped=True)
        return self.out6


class FusedPointStep:
    """Pointwise counterpart of FusedBPRStep: rows (user, item, label) -- recbole's pointwise layout, sampled negatives stacked
    behind the positives with label 0 -- MSE on the raw dot (EMCDR's default MF latent factor model, emcdr.py:111-122) or BCE
    on sigmoid(dot), plus ``reg_weight * EmbLoss(u_rows, i_rows)``; both tables updated row-wise on the batch's rows."""

    def __init__(self, user_table, item_table, max_batch, loss='mse', opt='adam', lr=1e-3, betas=(0.9, 0.999), eps=1e-8,
                 weight_decay=0.0, reg_weight=0.0, user_state=None, item_state=None, fuse_singles=True):
        assert user_table.is_cuda and item_table.is_cuda, 'FusedPointStep needs ROCm device tensors'
        assert user_table.shape[1] == item_table.shape[1]
        self.U, self.I = user_table, item_table
        self.D = user_table.shape[1]
        self.kind = B_.CDR_LOSS_MSE if loss == 'mse' else B_.CDR_LOSS_BCE
        self.opt = OPT_ADAM if opt == 'adam' else OPT_SGD
        self.lr, self.betas, self.eps, self.wd = lr, betas, eps, weight_decay
        self.reg_weight = reg_weight
        self.ustate = user_state if user_state is not None else RowwiseState(user_table, self.opt)
        self.istate = item_state if item_state is not None else RowwiseState(item_table, self.opt)
        dev = user_table.device
        Bm = int(max_batch)
        self.max_batch = Bm
        self.GU = torch.empty(Bm, self.D, device=dev, dtype=torch.float32)
        self.GI = torch.empty(Bm, self.D, device=dev, dtype=torch.float32)
        self.out6 = torch.zeros(12, device=dev, dtype=torch.float32)
        self.keys = torch.empty(2 * Bm, device=dev, dtype=torch.int32)        # one sort for both tables (see FusedBPRStep)
        self.perm = torch.empty(2 * Bm, device=dev, dtype=torch.int32)
        rows = max(user_table.shape[0], item_table.shape[0])
        need = ctypes.c_size_t(0)
        B_._check(B_.load().cdr_sort_workspace_bytes(2 * Bm, 2 << (rows - 1).bit_length(), ctypes.byref(need)),
                  'cdr_sort_workspace_bytes')
        self.ws = torch.empty(int(need.value), device=dev, dtype=torch.uint8)
        self._key_base = ctypes.c_uint32(0)
        # round 5: rows that occur once in the batch are updated by the forward kernel itself (cdr_point_step_fused), as in FusedBPRStep;
        # fuse_singles=False (or CDR_FUSE_SINGLES=0) keeps the two-pass form (the dimension-sharded step drives its halves)
        self.fuse_singles = bool(fuse_singles) and self.D % 4 == 0 and self.D <= 256 and os.environ.get('CDR_FUSE_SINGLES', '1') != '0'
        if self.fuse_singles:
            words = ctypes.c_int64(0)
            B_._check(B_.load().cdr_bpr_step_fused_heads_words(Bm, ctypes.byref(words)), 'cdr_bpr_step_fused_heads_words')
            self.flags = torch.zeros(4 * Bm, device=dev, dtype=torch.uint8)      # {user, item, -, -} per row
            self.heads = torch.empty(int(words.value), device=dev, dtype=torch.int32)

    def step(self, uid, iid, label):
        """uid / iid int64 [B], label fp32 [B].  Returns out6 (view; [0] = total loss)."""
        B = uid.numel()
        assert B <= self.max_batch
        if self.fuse_singles:
            us, its = self.ustate, self.istate
            us.advance(); its.advance()
            B_.call('cdr_point_step_fused', B_.ctx(self.U.device), B_.stream(), self.kind, self.opt, B_.f32(us.table), B_.f32(us.exp_avg),
                    B_.f32(us.exp_avg_sq), us.table.shape[0], B_.f32(its.table), B_.f32(its.exp_avg), B_.f32(its.exp_avg_sq), its.table.shape[0],
                    self.D, B_.i64(uid), B_.i64(iid), B_.f32(label), B, float(self.reg_weight), float(self.lr), float(self.betas[0]),
                    float(self.betas[1]), float(self.eps), float(self.wd), us.step, its.step, B_.f32(self.out6), B_.f32(self.GU), B_.f32(self.GI),
                    B_.raw(self.keys), B_.raw(self.perm), B_.raw(self.flags), B_.raw(self.heads), B_.raw(self.ws), self.ws.numel())
            return self.out6
        B_.call('cdr_point_fwd_grad', B_.ctx(self.U.device), B_.stream(), self.kind, B_.f32(self.U), B_.f32(self.I), self.D, B_.i64(uid),
                B_.i64(iid), B_.f32(label), B, float(self.reg_weight), B_.f32(self.out6), B_.f32(self.GU), B_.f32(self.GI))
        return self.sort_apply(uid, iid)

    def sort_apply(self, uid, iid):
        """Second half of the step (GU / GI / out6[4:6] in place): one sort for both tables, two row-wise applies."""
        self.sort_ids(uid, iid)
        return self.apply_sorted(uid.numel())

    def sort_ids(self, uid, iid):
        B = uid.numel()
        B_.call('cdr_sort_ids_two_tables', B_.ctx(self.U.device), B_.stream(), B_.i64(uid), B, self.U.shape[0], B_.i64(iid), B, None, 0,
                self.I.shape[0], B_.raw(self.keys), B_.raw(self.perm), ctypes.byref(self._key_base), B_.raw(self.ws), self.ws.numel())

    def apply_sorted(self, B):
        s = B_.stream()
        ctxh = B_.ctx(self.U.device)
        for st, lo, G, coef, base in ((self.ustate, 0, self.GU, self.out6[4:5], 0),
                                      (self.istate, B, self.GI, self.out6[5:6], self._key_base.value)):
            st.advance()
            B_.call('cdr_rowwise_apply', ctxh, s, self.opt, B_.f32(st.table), B_.f32(st.exp_avg), B_.f32(st.exp_avg_sq), self.D,
                    B_.raw(self.keys[lo:lo + B]), B_.raw(self.perm[lo:lo + B]), B, B_.f32(G), B, B, B_.f32(coef), float(self.lr),
                    float(self.betas[0]), float(self.betas[1]), float(self.eps), float(self.wd), st.step, None, int(base))
        return self.out6


class KMajorPointStep:
    """``FusedPointStep`` cut along recbole's pointwise batch layout: S positives, the user column tiled 1 + k times, items =
    [positives | k-major negatives], labels [1] * S + [0] * S k (TrainDataLoader._neg_sampling).  One lane group per POSITIVE gathers
    the user row once and sums its gradient over the 1 + k rows in registers; users that occur in one positive and item rows that occur
    once are updated by that kernel, the rest by the segmented applies (csrc/cdr_step.hip: point_fwd_apply_kmajor_kernel).  Same loss
    and per-row gradients as ``FusedPointStep`` on the S (1 + k) rows."""

    def __init__(self, user_table, item_table, max_positives, k=1, loss='mse', opt='adam', lr=1e-3, betas=(0.9, 0.999), eps=1e-8,
                 weight_decay=0.0, reg_weight=0.0, user_state=None, item_state=None):
        assert user_table.is_cuda and item_table.is_cuda, 'KMajorPointStep needs ROCm device tensors'
        assert user_table.shape[1] == item_table.shape[1] and user_table.shape[1] % 4 == 0 and user_table.shape[1] <= 256 and 1 <= int(k) <= 64
        self.U, self.I = user_table, item_table
        self.D, self.k = user_table.shape[1], int(k)
        self.kind = B_.CDR_LOSS_MSE if loss == 'mse' else B_.CDR_LOSS_BCE
        self.opt = OPT_ADAM if opt == 'adam' else OPT_SGD
        self.lr, self.betas, self.eps, self.wd, self.reg_weight = lr, betas, eps, weight_decay, reg_weight
        self.ustate = user_state if user_state is not None else RowwiseState(user_table, self.opt)
        self.istate = item_state if item_state is not None else RowwiseState(item_table, self.opt)
        dev = user_table.device
        Sm = int(max_positives)
        nI = Sm * (1 + self.k)
        self.max_positives = Sm
        self.max_batch = nI
        self.GU = torch.empty(Sm, self.D, device=dev, dtype=torch.float32)
        self.GI = torch.empty(nI, self.D, device=dev, dtype=torch.float32)
        fb, hw = ctypes.c_int64(0), ctypes.c_int64(0)
        B_._check(B_.load().cdr_bpr_step_fused_kmajor_sizes(Sm, self.k, ctypes.byref(fb), ctypes.byref(hw)), 'cdr_bpr_step_fused_kmajor_sizes')
        self.flags = torch.zeros(int(fb.value), device=dev, dtype=torch.uint8)
        self.heads = torch.empty(int(hw.value), device=dev, dtype=torch.int32)
        self.out6 = torch.zeros(12, device=dev, dtype=torch.float32)
        self.keys = torch.empty(Sm + nI, device=dev, dtype=torch.int32)
        self.perm = torch.empty(Sm + nI, device=dev, dtype=torch.int32)
        rows = max(user_table.shape[0], item_table.shape[0])
        need = ctypes.c_size_t(0)
        B_._check(B_.load().cdr_sort_workspace_bytes(Sm + nI, 2 << (rows - 1).bit_length(), ctypes.byref(need)), 'cdr_sort_workspace_bytes')
        self.ws = torch.empty(int(need.value), device=dev, dtype=torch.uint8)

    def step(self, uid, iid, label):
        """uid int64 [S (1 + k)] tiled (or [S]: the first S entries are read), iid int64 [S (1 + k)], label fp32 [S (1 + k)].
        Returns out6 (view; [0] = total loss)."""
        n = iid.numel()
        S = n // (1 + self.k)
        assert S * (1 + self.k) == n and S <= self.max_positives and uid.numel() >= S and label.numel() == n
        us, its = self.ustate, self.istate
        us.advance(); its.advance()
        B_.call('cdr_point_step_fused_kmajor', B_.ctx(self.U.device), B_.stream(), self.kind, self.opt, B_.f32(us.table), B_.f32(us.exp_avg),
                B_.f32(us.exp_avg_sq), us.table.shape[0], B_.f32(its.table), B_.f32(its.exp_avg), B_.f32(its.exp_avg_sq), its.table.shape[0],
                self.D, B_.i64(uid), B_.i64(iid), B_.f32(label), S, self.k, float(self.reg_weight), float(self.lr), float(self.betas[0]),
                float(self.betas[1]), float(self.eps), float(self.wd), us.step, its.step, B_.f32(self.out6), B_.f32(self.GU), B_.f32(self.GI),
                B_.raw(self.keys), B_.raw(self.perm), B_.raw(self.flags), B_.raw(self.heads), B_.raw(self.ws), self.ws.numel())
        return self.out6


class FusedMapStep:
    """EMCDR's OVERLAP phase (emcdr.py:133-137 ``calculate_map_loss``: MSE(mapping(source_e[idx]), target_e[idx])) as an
    O(batch) step: the two embedding tables are updated row-wise on the overlapped ids only; the mapping function's own
    parameters (a few hundred KB) keep the exact dense Adam.

    ``mapping_fn``  x [n, Ds] -> [n, Dt] built from functional.linear (autograd through the native GEMM kernels), e.g. a
                    model's ``apply_mapping``;  ``mapping_params`` its parameters.
    ``group``       optional process group: the tables are then this rank's row shards (row r % G == rank at r // G) and
                    ``step`` takes GLOBAL ids.  Both tables use the same sharding rule and an overlapped id names the same
                    entity in both, so after ONE all-to-all of ids everything is local: the "overlapped-user transfer
                    step" moves 8 B per id, never a row.  The MSE mean is over the global batch; the mapping gradients
                    are all-reduced (every rank then takes the same dense Adam step on its replica).
    """

    def __init__(self, source_table, target_table, mapping_fn, mapping_params, max_batch, opt='adam', lr=1e-3,
                 betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, group=None, source_state=None, target_state=None, layers=None):
        from .trainer.trainer import DenseAdam
        assert source_table.is_cuda and target_table.is_cuda, 'FusedMapStep needs ROCm device tensors'
        self.S, self.T = source_table, target_table
        self.mapping_fn = mapping_fn
        self.mapping_params = list(mapping_params)
        self.opt = OPT_ADAM if opt == 'adam' else OPT_SGD
        self.lr, self.betas, self.eps, self.wd = lr, betas, eps, weight_decay
        # the row-wise moments may be shared with the FusedBPRStep objects of the SOURCE / TARGET phases (one optimizer
        # state per table across phases, like the reference's single Adam instance: trainer.py:30-41)
        self.sstate = source_state if source_state is not None else RowwiseState(source_table, self.opt)
        self.tstate = target_state if target_state is not None else RowwiseState(target_table, self.opt)
        self.map_opt = DenseAdam(self.mapping_params, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay) \
            if self.opt == OPT_ADAM else None
        self.group = group
        self.loss = torch.zeros((), device=source_table.device, dtype=torch.float32)
        self._ws = None
        # ``layers``: the mapping function's structure [(weight [out, in], bias or None, ACT_NONE | ACT_TANH), ...] (emcdr.py:59-64,
        # 86-93).  With it, batches of DISTINCT ids -- what the reference's OverlapDataloader yields -- run as two native launches
        # (csrc/cdr_mapstep.hip): ``step(idx, unique=True)``.
        self.layers = None
        self._uws = None
        if layers is not None and group is None:
            dims = [layers[0][0].shape[1]] + [w.shape[0] for w, _, _ in layers]
            need = ctypes.c_size_t(0)
            L = len(layers)
            if L <= 4 and B_.load().cdr_map_step_plan(L, (ctypes.c_int * (L + 1))(*dims), (ctypes.c_int * L)(*[int(b is not None) for _, b, _ in layers]),
                                                       max(int(max_batch), 1), ctypes.byref(need)) == 0:
                self.layers, self._dims = list(layers), dims
                self._loss1 = torch.zeros(1, device=source_table.device, dtype=torch.float32)

    def _step_unique(self, idx):
        """Two launches: gather + mapping + MSE + backward + in-place row updates + gradient partials, then reduction + dense Adam on
        the mapping.  Update counts are device counters (capturable)."""
        n = idx.numel()
        dev = self.S.device
        L = len(self.layers)
        need = ctypes.c_size_t(0)
        dims_c = (ctypes.c_int * (L + 1))(*self._dims)
        B_._check(B_.load().cdr_map_step_plan(L, dims_c, (ctypes.c_int * L)(*[int(b is not None) for _, b, _ in self.layers]), n,
                                              ctypes.byref(need)), 'cdr_map_step_plan')
        if self._uws is None or self._uws.numel() < need.value:
            self._uws = torch.empty(int(need.value), device=dev, dtype=torch.uint8)
        adam = self.opt == OPT_ADAM
        ptrs = lambda xs: (ctypes.c_void_p * L)(*[None if x is None else x.data_ptr() for x in xs])
        Ws = [w.data for w, _, _ in self.layers]
        bs = [None if b is None else b.data for _, b, _ in self.layers]
        mW = vW = mb = vb = sW = sb = [None] * L
        if adam:
            def st(p):
                d = self.map_opt.state[p]
                if not d:
                    d['step'] = torch.zeros(1, device=dev, dtype=torch.int64)
                    d['exp_avg'] = torch.zeros_like(p); d['exp_avg_sq'] = torch.zeros_like(p)
                return d
            sts = [st(w) for w, _, _ in self.layers]
            stb = [None if b is None else st(b) for _, b, _ in self.layers]
            mW, vW, sW = [d['exp_avg'] for d in sts], [d['exp_avg_sq'] for d in sts], [d['step'] for d in sts]
            mb = [None if d is None else d['exp_avg'] for d in stb]; vb = [None if d is None else d['exp_avg_sq'] for d in stb]
            sb = [None if d is None else d['step'] for d in stb]
        keep = [Ws, bs, mW, vW, mb, vb, sW, sb]                         # alive until the call is enqueued
        ss, ts_ = (self.sstate.step_dev, self.tstate.step_dev) if adam else (None, None)
        B_.call('cdr_map_step_unique', B_.ctx(dev), B_.stream(), self.opt, B_.f32(self.S), B_.f32(self.sstate.exp_avg),
                B_.f32(self.sstate.exp_avg_sq), B_.f32(self.T), B_.f32(self.tstate.exp_avg), B_.f32(self.tstate.exp_avg_sq), B_.i64(idx), n,
                L, dims_c, (ctypes.c_int * L)(*[int(a) for _, _, a in self.layers]), ptrs(Ws), ptrs(bs), ptrs(mW), ptrs(vW), ptrs(mb),
                ptrs(vb), ptrs(sW), ptrs(sb), B_.i64(ss), B_.i64(ts_), float(self.lr), float(self.betas[0]), float(self.betas[1]),
                float(self.eps), float(self.wd), B_.f32(self._loss1), B_.raw(self._uws), self._uws.numel())
        del keep
        if not torch.cuda.is_current_stream_capturing():       # (a capture enqueues nothing: the host counts advance per replay)
            self.sstate.advance(device_bumped=True)
            self.tstate.advance(device_bumped=True)
        self.loss = self._loss1[0]
        return self.loss

    # ---- hipGraph replay of the two-launch step (the reference's default overlap_batch_size is 100: launch-bound) ----------------
    def capture(self, OB):
        """Capture the distinct-id step for batches of exactly OB ids.  ``replay(idx)`` then costs one id copy + one graph launch;
        every update count the kernels read lives on the device."""
        assert self.layers is not None and self.group is None
        dev = self.S.device
        self._OB = OB
        self._idx = torch.arange(1, OB + 1, device=dev, dtype=torch.int64)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            B_.ctx(dev)                                   # the native context of the capture stream must exist beforehand
            if self.opt == OPT_ADAM:
                self.sstate.step_dev, self.tstate.step_dev
                for w, b, _ in self.layers:                # the mapping's Adam state and the workspace: created outside the capture
                    for prm in (w, b):
                        if prm is not None and not self.map_opt.state[prm]:
                            d = self.map_opt.state[prm]
                            d['step'] = torch.zeros(1, device=dev, dtype=torch.int64)
                            d['exp_avg'] = torch.zeros_like(prm); d['exp_avg_sq'] = torch.zeros_like(prm)
            need = ctypes.c_size_t(0)
            L = len(self.layers)
            B_._check(B_.load().cdr_map_step_plan(L, (ctypes.c_int * (L + 1))(*self._dims),
                                                  (ctypes.c_int * L)(*[int(b is not None) for _, b, _ in self.layers]), OB,
                                                  ctypes.byref(need)), 'cdr_map_step_plan')
            if self._uws is None or self._uws.numel() < need.value:
                self._uws = torch.empty(int(need.value), device=dev, dtype=torch.uint8)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        host = (self.sstate._step, self.tstate._step)
        self._graph = torch.cuda.CUDAGraph()
        with B_.capturing(self._graph, side):
            self._step_unique(self._idx)
        self.sstate._step, self.tstate._step = host       # the capture enqueued nothing: host counts advance per replay
        return self

    def replay(self, idx=None):
        if idx is not None:
            self._idx.copy_(idx.reshape(-1)[:self._OB])
        self._graph.replay()
        self.sstate.advance(device_bumped=True)
        self.tstate.advance(device_bumped=True)
        self.loss = self._loss1[0]
        return self.loss

    def _route(self, idx):
        """Global ids -> the local row indices this rank owns (one all-to-all of ids), plus the global batch size."""
        import torch.distributed as dist
        from .shard import NativeOps, _a2a, _gather_counts
        G, rank = dist.get_world_size(self.group), dist.get_rank(self.group)
        ops = NativeOps(idx.device) if not hasattr(self, '_ops') else self._ops
        self._ops = ops
        perm, counts = ops.route(idx, None, G)
        send = ops.permute(idx, None, perm, G)
        allc = torch.stack(_gather_counts(counts, idx.numel(), self.group, G)).tolist()          # one host sync
        t_send = [int(c) for c in allc[rank][:G]]
        t_recv = [int(allc[r][rank]) for r in range(G)]
        n_global = sum(int(allc[r][G]) for r in range(G))
        if n_global == 0:                          # nothing anywhere: no zero-byte all-to-all either
            return send, 0
        return _a2a(send, t_send, t_recv, self.group), n_global

    def step(self, idx, unique=False):
        """idx: int64 device tensor of overlapped ids, any shape ([OB,1] from the OverlapDataloader).  Returns the loss
        (device scalar; the global-batch MSE when sharded).  ``unique=True`` asserts that the ids are pairwise distinct (slices of
        a permutation, as the reference's loader yields them): the two-launch path of csrc/cdr_mapstep.hip."""
        idx = idx.reshape(-1).contiguous()
        n_global = idx.numel()
        if unique and self.layers is not None and n_global > 0:
            return self._step_unique(idx)
        if self.group is not None:
            idx, n_global = self._route(idx)
        if n_global == 0:                          # an empty batch (on every rank) is a no-op: no state advances
            self.loss = torch.zeros((), device=self.S.device, dtype=torch.float32)
            return self.loss
        n = idx.numel()
        D_s, D_t = self.S.shape[1], self.T.shape[1]
        dev = self.S.device
        s = B_.stream()
        for p in self.mapping_params:
            p.grad = None
        if n:
            src = torch.empty(n, D_s, device=dev, dtype=torch.float32)
            tgt = torch.empty(n, D_t, device=dev, dtype=torch.float32)
            B_.call('cdr_gather_rows', s, B_.f32(self.S), D_s, B_.i64(idx), n, B_.f32(src))
            B_.call('cdr_gather_rows', s, B_.f32(self.T), D_t, B_.i64(idx), n, B_.f32(tgt))
            src.requires_grad_(True); tgt.requires_grad_(True)
            from . import functional as F_
            local = F_.mse_loss(self.mapping_fn(src), tgt)              # mean over n * Dt
            loss = local * (float(n) / float(n_global))                  # this rank's share of the global mean
            loss.backward()
            loss = loss.detach()
        else:
            loss = torch.zeros((), device=dev, dtype=torch.float32)
        if self.group is not None:
            import torch.distributed as dist
            flat = torch.cat([loss.reshape(1)] + [(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1)
                                                  for p in self.mapping_params])
            dist.all_reduce(flat, group=self.group)
            loss, off = flat[0], 1
            for p in self.mapping_params:
                p.grad = flat[off:off + p.numel()].view_as(p).clone()
                off += p.numel()
        self.loss = loss
        self.sstate.advance()                  # per TABLE, also on a rank that received no ids (the shards stay in step)
        self.tstate.advance()
        if n:
            ctxh = B_.ctx(dev)
            need = ctypes.c_size_t(0)
            B_._check(B_.load().cdr_sort_workspace_bytes(n, self.S.shape[0], ctypes.byref(need)), 'cdr_sort_workspace_bytes')
            if self._ws is None or self._ws.numel() < need.value:
                self._ws = torch.empty(int(need.value), device=dev, dtype=torch.uint8)
            keys = torch.empty(n, device=dev, dtype=torch.int32)
            perm = torch.empty(n, device=dev, dtype=torch.int32)
            # the same ids index both tables: one sort serves both applies
            B_.call('cdr_sort_ids', ctxh, s, B_.i64(idx), n, None, 0, self.S.shape[0], B_.raw(keys), B_.raw(perm),
                    B_.raw(self._ws), self._ws.numel())
            for st, g in ((self.sstate, src.grad), (self.tstate, tgt.grad)):
                B_.call('cdr_rowwise_apply', ctxh, s, self.opt, B_.f32(st.table), B_.f32(st.exp_avg), B_.f32(st.exp_avg_sq),
                        st.table.shape[1], B_.raw(keys), B_.raw(perm), n, B_.f32(g), n, 0, None, float(self.lr),
                        float(self.betas[0]), float(self.betas[1]), float(self.eps), float(self.wd), st.step, None, 0)
        if self.map_opt is not None:
            self.map_opt.step()
        else:
            with torch.no_grad():
                for p in self.mapping_params:
                    if p.grad is not None:
                        p.add_(p.grad + self.wd * p if self.wd else p.grad, alpha=-self.lr)
        return self.loss
