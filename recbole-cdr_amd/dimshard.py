"""BPR training step over DIMENSION-sharded embedding tables.

Rank r holds columns ``[r*Ds, (r+1)*Ds)`` of every row of the user and the item table (``Ds = D / world``) with the
matching slice of the row-wise optimizer state.  Every rank brings its own batch (weak scaling, as shard.ShardedBPRStep);
the step is

    all-gather  the triples' ids, narrowed to int32       12 B per triple (cdr_ids_pack32 / cdr_ids_unpack32)
    cdr_bpr_partial_diff   <u,p> - <u,n> over my columns   (csrc/cdr_dimshard.hip)
    all-reduce  one float per triple + the two EmbLoss norms        || cdr_sort_ids_two_tables (needs only the ids)
    cdr_bpr_grad_from_diff + 2 x cdr_rowwise_apply                  (the single-GPU fused step on [rows, Ds])

-- 16 B per triple on xGMI against the ~2.1 KB of the row exchange (emcdr.py:98-108,119-131 need only the dot products to
cross the column cut; every gradient element stays with the rank that holds its column).  No bucket counts, no host
sync: all sizes are static.  The price is that every rank walks the GLOBAL batch on rows 1/world as wide, so the ids are
sorted ``world`` times over and gathers shrink to ``4*Ds`` bytes.

The batch size must be the same on every rank (a sharded loader pads or drops the ragged tail).

Contents: the layout transposes (``dim_shard_of``, ``dim_to_row_shards`` / ``row_to_dim_shards`` between all ranks,
``cols_to_row_shards`` when one domain's tables live on half of the ranks -- bench.py's domain groups), their RowwiseState forms,
``ShardedTables`` (a model's tables in whichever layout the phase wants), and the steps ``DimShardedBPRStep`` /
``DimShardedPointStep`` with ``step(..., next_batch=)`` prefetching the next id exchange."""
import contextlib

import torch
import torch.distributed as dist

OPT_SGD, OPT_ADAM = 0, 1


def dim_shard_of(full_table, world, rank):
    """Columns of ``full_table`` [rows, D] that rank ``rank`` of ``world`` holds -> contiguous [rows, D / world]."""
    D = full_table.shape[1]
    assert D % (4 * world) == 0, 'D / world must be a multiple of 4 (one float4 per lane)'
    Ds = D // world
    return full_table[:, rank * Ds:(rank + 1) * Ds].contiguous()


def dim_to_row_shards(cols, group=None, force=False):
    """[rows, Ds] column slice -> this rank's ROW shard [ceil-ish(rows / G), D] (row r % G == rank at r // G): the layout
    shard.ShardedFullSort evaluates on.  One all-to-all of the whole slice (rows * Ds * 4 B per rank), once per evaluation."""
    G, rank = dist.get_world_size(group), dist.get_rank(group)
    if G == 1 and not force:                  # force: issue the (identity) all-to-all anyway -- the one-GPU RCCL test
        return cols
    rows, Ds = cols.shape
    from .shard import shard_rows
    n_q = [shard_rows(rows, G, q) for q in range(G)]
    send = torch.cat([cols[q::G] for q in range(G)])                       # [rows, Ds], grouped by destination
    recv = torch.empty(G * n_q[rank], Ds, device=cols.device, dtype=cols.dtype)
    dist.all_to_all_single(recv, send, output_split_sizes=[n_q[rank]] * G, input_split_sizes=n_q, group=group)
    return recv.view(G, n_q[rank], Ds).permute(1, 0, 2).reshape(n_q[rank], G * Ds).contiguous()


def cols_to_row_shards(cols, rows, Ds, holders, group=None):
    """The transpose when only SOME ranks hold the table (one domain's tables live on that domain's half of the node):
    ``holders`` = ascending ranks of ``group`` that hold column blocks 0, 1, ... ([rows, Ds] each; ``cols`` is None elsewhere)
    -> on EVERY rank of ``group`` its row shard [rows_r, len(holders) * Ds].  One all-to-all; non-holders send nothing."""
    from .shard import shard_rows
    W, rank = dist.get_world_size(group), dist.get_rank(group)
    holders = list(holders)
    assert holders == sorted(holders) and (cols is not None) == (rank in holders)
    n_q = [shard_rows(rows, W, q) for q in range(W)]
    if cols is not None:
        send, in_splits = torch.cat([cols[q::W] for q in range(W)]), n_q
        dev = cols.device
    else:
        dev = torch.device('cuda', torch.cuda.current_device()) if torch.cuda.is_available() else torch.device('cpu')
        send, in_splits = torch.empty(0, Ds, device=dev, dtype=torch.float32), [0] * W
    out_splits = [n_q[rank] if q in holders else 0 for q in range(W)]
    recv = torch.empty(len(holders) * n_q[rank], Ds, device=dev, dtype=torch.float32)
    dist.all_to_all_single(recv, send, output_split_sizes=out_splits, input_split_sizes=in_splits, group=group)
    return recv.view(len(holders), n_q[rank], Ds).permute(1, 0, 2).reshape(n_q[rank], len(holders) * Ds).contiguous()


def state_cols_to_row_shards(state, rows, Ds, holders, adam, group=None):
    """``cols_to_row_shards`` for a table with its row-wise optimizer state (``state`` is None on non-holders; the update count is
    taken from the first holder)."""
    from .fused import RowwiseState
    out = RowwiseState.__new__(RowwiseState)
    dev = state.table.device if state is not None else torch.device('cuda', torch.cuda.current_device()) if torch.cuda.is_available() else torch.device('cpu')
    cnt = torch.tensor([state.step if state is not None else 0], device=dev, dtype=torch.int64)
    dist.broadcast(cnt, dist.get_global_rank(group, holders[0]) if group is not None else holders[0], group=group)
    out.step = int(cnt.item())
    for name in ('table', 'exp_avg', 'exp_avg_sq'):
        if name != 'table' and not adam:
            setattr(out, name, None)
            continue
        t = getattr(state, name) if state is not None else None
        setattr(out, name, cols_to_row_shards(t, rows, Ds, holders, group))
        if state is not None:
            setattr(state, name, None)                    # consumed: peak memory = one tensor extra
        del t
    return out


def row_shards_to_cols(row_shard, rows, D, holders, group=None):
    """Inverse of ``cols_to_row_shards``: every rank's row shard [rows_r, D] -> column block j [rows, D / len(holders)] on
    ``holders[j]``, None on the other ranks (they send, and receive nothing)."""
    from .shard import shard_rows
    W, rank = dist.get_world_size(group), dist.get_rank(group)
    holders = list(holders)
    h = len(holders)
    Ds = D // h
    n_q = [shard_rows(rows, W, q) for q in range(W)]
    blocks = row_shard.view(n_q[rank], h, Ds).permute(1, 0, 2).contiguous()                 # [h, rows_r, Ds], ascending holder rank
    in_splits = [n_q[rank] if q in holders else 0 for q in range(W)]
    mine = rank in holders
    out_splits = n_q if mine else [0] * W
    recv = torch.empty(rows if mine else 0, Ds, device=row_shard.device, dtype=row_shard.dtype)
    dist.all_to_all_single(recv, blocks.view(h * n_q[rank], Ds), output_split_sizes=out_splits, input_split_sizes=in_splits, group=group)
    if not mine:
        return None
    out = torch.empty(rows, Ds, device=row_shard.device, dtype=row_shard.dtype)
    o = 0
    for q in range(W):
        out[q::W] = recv[o:o + n_q[q]]
        o += n_q[q]
    return out


def state_row_shards_to_cols(state, rows, D, holders, group=None):
    """``row_shards_to_cols`` for a table with its moments: a RowwiseState on the holders, None elsewhere."""
    from .fused import RowwiseState
    parts = {}
    for name in ('table', 'exp_avg', 'exp_avg_sq'):
        t = getattr(state, name)
        parts[name] = row_shards_to_cols(t, rows, D, holders, group) if t is not None else None
        setattr(state, name, None)
    if dist.get_rank(group) not in list(holders):
        return None
    out = RowwiseState.__new__(RowwiseState)
    out.step = state.step
    out.table, out.exp_avg, out.exp_avg_sq = parts['table'], parts['exp_avg'], parts['exp_avg_sq']
    return out


def row_to_dim_shards(row_shard, total_rows, group=None, force=False):
    """Inverse of ``dim_to_row_shards``: this rank's row shard [rows_r, D] -> its column slice [total_rows, D / G]."""
    G, rank = dist.get_world_size(group), dist.get_rank(group)
    if G == 1 and not force:
        return row_shard
    from .shard import shard_rows
    n_q = [shard_rows(total_rows, G, q) for q in range(G)]
    D = row_shard.shape[1]
    Ds = D // G
    send = row_shard.view(n_q[rank], G, Ds).permute(1, 0, 2).contiguous().view(G * n_q[rank], Ds)
    recv = torch.empty(total_rows, Ds, device=row_shard.device, dtype=row_shard.dtype)
    dist.all_to_all_single(recv, send, output_split_sizes=n_q, input_split_sizes=[n_q[rank]] * G, group=group)
    out = torch.empty(total_rows, Ds, device=row_shard.device, dtype=row_shard.dtype)
    o = 0
    for q in range(G):
        out[q::G] = recv[o:o + n_q[q]]
        o += n_q[q]
    return out


def state_to_row_shards(state, group=None, consume=False):
    """fused.RowwiseState of a column slice -> RowwiseState of this rank's row shard (table and both moments re-laid out, the
    update count kept): the phase switch from dimension-sharded BPR epochs to the row-sharded OVERLAP step / full-sort.
    ``consume``: drop each column-slice tensor from ``state`` as soon as it is converted (peak memory = one tensor extra)."""
    from .fused import RowwiseState
    out = RowwiseState.__new__(RowwiseState)
    out.step = state.step
    for name in ('table', 'exp_avg', 'exp_avg_sq'):
        t = getattr(state, name)
        setattr(out, name, dim_to_row_shards(t, group) if t is not None else None)
        if consume:
            setattr(state, name, None)
        del t
    return out


def state_to_dim_shards(state, total_rows, group=None):
    """Inverse of ``state_to_row_shards`` (OVERLAP epochs done, back to dimension-sharded BPR epochs)."""
    from .fused import RowwiseState
    out = RowwiseState.__new__(RowwiseState)
    out.step = state.step
    for name in ('table', 'exp_avg', 'exp_avg_sq'):
        t = getattr(state, name)
        setattr(out, name, row_to_dim_shards(t, total_rows, group) if t is not None else None)
    return out


class ShardedTables:
    """The embedding tables of one model over the ranks of ``group``, each with its row-wise optimizer state, in whichever of
    the two layouts the current phase wants: 'dim' (column blocks of every row on the table's ``holders`` -- all ranks, or one
    half of them when each domain trains on its own half: BPR / MF steps) or 'row' (rows r % G == rank on ALL ranks: the OVERLAP
    step, full-sort evaluation).  ``state(name, layout)`` transposes table and moments when the layout changes (update count
    kept; None on a rank that holds nothing of the table); ``rows(name)`` is a row-shard view for evaluation that leaves a
    'dim' state where it is (the table alone is transposed into a temporary that ``touched(name)`` -- called by every training
    step -- invalidates)."""

    def __init__(self, group, opt_code):
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.opt = opt_code
        self.entries = {}

    def adopt(self, name, full_table, layout='dim', holders=None, hgroup=None):
        """``full_table``: the replicated [rows, D] tensor (identical on every rank) -> this rank's shard (None if it holds
        nothing) + a fresh state.  ``holders`` / ``hgroup``: the ranks of ``group`` (ascending) that hold the 'dim' layout's
        column blocks and their process group; default all ranks."""
        from .fused import RowwiseState
        from .shard import shard_of
        holders = list(range(self.world)) if holders is None else list(holders)
        if layout == 'dim':
            shard = dim_shard_of(full_table, len(holders), holders.index(self.rank)) if self.rank in holders else None
        else:
            shard = shard_of(full_table, self.world, self.rank).contiguous()
        self.entries[name] = {'layout': layout, 'state': RowwiseState(shard, self.opt) if shard is not None else None,
                              'rows': full_table.shape[0], 'D': full_table.shape[1], 'version': 0, 'eval_rows': None,
                              'holders': holders, 'hgroup': hgroup if hgroup is not None else self.group,
                              'device': full_table.device}
        return shard

    def layout(self, name):
        return self.entries[name]['layout']

    def version(self, name):
        return self.entries[name]['version']

    def holder_group(self, name):
        return self.entries[name]['hgroup']

    def state(self, name, layout):
        e = self.entries[name]
        if e['layout'] != layout:
            h = len(e['holders'])
            if layout == 'row':
                e['state'] = state_cols_to_row_shards(e['state'], e['rows'], e['D'] // h, e['holders'], self.opt == OPT_ADAM, self.group)
            else:
                e['state'] = state_row_shards_to_cols(e['state'], e['rows'], e['D'], e['holders'], self.group)
            e['layout'], e['eval_rows'] = layout, None
            e['version'] += 1
        return e['state']

    def touched(self, name):
        self.entries[name]['eval_rows'] = None

    def rows(self, name):
        e = self.entries[name]
        if e['layout'] == 'row':
            return e['state'].table
        if e['eval_rows'] is None:
            t = e['state'].table if e['state'] is not None else None
            e['eval_rows'] = cols_to_row_shards(t, e['rows'], e['D'] // len(e['holders']), e['holders'], self.group)
        return e['eval_rows']

    def full(self, name):
        """The replicated [rows, D] table again (checkpointing / hand-over to a single-process run)."""
        from .shard import shard_rows
        e = self.entries[name]
        t = e['state'].table if e['state'] is not None else None
        src = lambda q: dist.get_global_rank(self.group, q) if self.group is not None else q
        if e['layout'] == 'dim':
            Ds = e['D'] // len(e['holders'])
            parts = []
            for q in e['holders']:
                part = t.clone() if q == self.rank else torch.empty(e['rows'], Ds, device=e['device'], dtype=torch.float32)
                dist.broadcast(part, src(q), group=self.group)
                parts.append(part)
            return torch.cat(parts, dim=1)
        out = torch.empty(e['rows'], e['D'], device=t.device, dtype=t.dtype)
        for q in range(self.world):
            part = torch.empty(shard_rows(e['rows'], self.world, q), e['D'], device=t.device, dtype=t.dtype)
            if q == self.rank:
                part.copy_(t)
            dist.broadcast(part, src(q), group=self.group)
            out[q::self.world] = part
        return out


class NativeDimOps:
    """libcdrhip arithmetic for DimShardedBPRStep: the two kernels of csrc/cdr_dimshard.hip around fused.FusedBPRStep's
    buffers, sort and row-wise applies (run on [rows, Ds] tables)."""

    def __init__(self, user_cols, item_cols, max_global_batch, **kw):
        from . import binding as B_
        from .fused import FusedBPRStep
        self.B_ = B_
        # the forward is cut in two around the all-reduce; since round 5 its second half is the one-GPU step's forward-and-update pass
        # (rows that occur once in the GLOBAL batch are updated by the pass that re-gathers them: cdr_bpr_step_from_diff).
        # fuse_singles=False (or CDR_FUSE_SINGLES=0) keeps the round-1 form: compact gradient rows for every triple + segmented applies.
        self.fs = FusedBPRStep(user_cols, item_cols, max_global_batch, fuse_singles=kw.pop('fuse_singles', True), device_counts=False, **kw)
        self.fused = self.fs.fuse_singles
        self.out = self.fs.out6

    def pack_ids(self, a, b, c, out32):
        """c: int64 ids (pairwise: negatives) or fp32 labels (pointwise rows)."""
        B_ = self.B_
        if not hasattr(self, '_bad'):
            self._bad = torch.zeros(1, device=a.device, dtype=torch.int32)
        is_label = c.dtype == torch.float32
        B_.call('cdr_ids_pack32', B_.stream(), B_.i64(a), B_.i64(b), None if is_label else B_.i64(c), B_.f32(c) if is_label else None,
                a.numel(), B_.raw(out32), B_.raw(self._bad))

    def unpack_ids(self, gathered32, world, Bl, out64, label_out=None):
        B_ = self.B_
        B_.call('cdr_ids_unpack32', B_.stream(), B_.raw(gathered32), int(world), int(Bl), B_.i64(out64),
                B_.f32(label_out) if label_out is not None else None)

    def ids_fit(self):
        """False if any id handed to ``pack_ids`` so far did not fit int32 (synchronises; the tables here never have 2^31 rows)."""
        return not hasattr(self, '_bad') or int(self._bad.item()) == 0

    def partial(self, uid, pid, nid, diff):
        B_, fs = self.B_, self.fs
        B_.call('cdr_bpr_partial_diff', B_.ctx(fs.U.device), B_.stream(), B_.f32(fs.U), B_.f32(fs.I), fs.D, B_.i64(uid),
                B_.i64(pid), B_.i64(nid), uid.numel(), B_.f32(diff))

    def grad_apply(self, uid, pid, nid, diff):
        B_, fs = self.B_, self.fs
        if self.fused:
            us, its = fs.ustate, fs.istate
            us.advance(); its.advance()
            B_.call('cdr_bpr_step_from_diff', B_.ctx(fs.U.device), B_.stream(), fs.opt, B_.f32(us.table), B_.f32(us.exp_avg), B_.f32(us.exp_avg_sq),
                    B_.f32(its.table), B_.f32(its.exp_avg), B_.f32(its.exp_avg_sq), fs.D, B_.i64(uid), B_.i64(pid), B_.i64(nid), uid.numel(),
                    float(fs.gamma), float(fs.reg_weight), float(fs.lr), float(fs.betas[0]), float(fs.betas[1]), float(fs.eps), float(fs.wd),
                    us.step, its.step, B_.f32(diff), int(fs._key_base.value), B_.f32(fs.out6), B_.f32(fs.GU), B_.f32(fs.GP), B_.raw(fs.keys),
                    B_.raw(fs.perm), B_.raw(fs.flags), B_.raw(fs.heads))
            return fs.out6
        B_.call('cdr_bpr_grad_from_diff', B_.ctx(fs.U.device), B_.stream(), B_.f32(fs.U), B_.f32(fs.I), fs.D, B_.i64(uid),
                B_.i64(pid), B_.i64(nid), uid.numel(), float(fs.gamma), float(fs.reg_weight), B_.f32(diff), B_.f32(fs.out6),
                B_.f32(fs.GU), B_.f32(fs.GP))
        return fs.apply_sorted(uid.numel())

    def presort(self, uid, pid, nid):
        """The id sort of the step (fused form: + the occurrence flags and duplicate-segment heads); needs only the ids, so the step issues
        it while the partial scores are being all-reduced."""
        if self.fused:
            import ctypes
            B_, fs = self.B_, self.fs
            B_.call('cdr_bpr_step_presort', B_.ctx(fs.U.device), B_.stream(), B_.i64(uid), B_.i64(pid), B_.i64(nid), uid.numel(), fs.U.shape[0],
                    fs.I.shape[0], B_.raw(fs.keys), B_.raw(fs.perm), B_.raw(fs.flags), B_.raw(fs.heads), B_.raw(fs.ws), fs.ws_bytes,
                    ctypes.byref(fs._key_base))
            return
        self.fs.sort_ids(uid, pid, nid)


class NativePointDimOps(NativeDimOps):
    """Pointwise rows (user, item, label): fused.FusedPointStep's buffers around cdr_point_partial_dot / cdr_point_grad_from_dot."""

    def __init__(self, user_cols, item_cols, max_global_batch, **kw):
        from . import binding as B_
        from .fused import FusedPointStep
        self.B_ = B_
        # the two halves around the all-reduce; since round 5 the second half is the forward-and-update pass fed with the given dots
        # (cdr_point_step_from_dot); fuse_singles=False keeps the two-pass form
        self.fs = FusedPointStep(user_cols, item_cols, max_global_batch, fuse_singles=kw.pop('fuse_singles', True), **kw)
        self.fused = self.fs.fuse_singles
        self.out = self.fs.out6

    def partial(self, uid, iid, label, dot):
        B_, fs = self.B_, self.fs
        B_.call('cdr_point_partial_dot', B_.ctx(fs.U.device), B_.stream(), B_.f32(fs.U), B_.f32(fs.I), fs.D, B_.i64(uid), B_.i64(iid),
                uid.numel(), B_.f32(dot))

    def grad_apply(self, uid, iid, label, dot):
        B_, fs = self.B_, self.fs
        if self.fused:
            us, its = fs.ustate, fs.istate
            us.advance(); its.advance()
            B_.call('cdr_point_step_from_dot', B_.ctx(fs.U.device), B_.stream(), fs.kind, fs.opt, B_.f32(us.table), B_.f32(us.exp_avg),
                    B_.f32(us.exp_avg_sq), B_.f32(its.table), B_.f32(its.exp_avg), B_.f32(its.exp_avg_sq), fs.D, B_.i64(uid), B_.i64(iid),
                    B_.f32(label), uid.numel(), float(fs.reg_weight), float(fs.lr), float(fs.betas[0]), float(fs.betas[1]), float(fs.eps),
                    float(fs.wd), us.step, its.step, B_.f32(dot), int(fs._key_base.value), B_.f32(fs.out6), B_.f32(fs.GU), B_.f32(fs.GI),
                    B_.raw(fs.keys), B_.raw(fs.perm), B_.raw(fs.flags), B_.raw(fs.heads))
            return fs.out6
        B_.call('cdr_point_grad_from_dot', B_.ctx(fs.U.device), B_.stream(), fs.kind, B_.f32(fs.U), B_.f32(fs.I), fs.D, B_.i64(uid),
                B_.i64(iid), B_.f32(label), uid.numel(), float(fs.reg_weight), B_.f32(dot), B_.f32(fs.out6), B_.f32(fs.GU), B_.f32(fs.GI))
        return fs.apply_sorted(uid.numel())

    def presort(self, uid, iid, label):
        if self.fused:
            import ctypes
            B_, fs = self.B_, self.fs
            B_.call('cdr_point_step_presort', B_.ctx(fs.U.device), B_.stream(), B_.i64(uid), B_.i64(iid), uid.numel(), fs.U.shape[0], fs.I.shape[0],
                    B_.raw(fs.keys), B_.raw(fs.perm), B_.raw(fs.flags), B_.raw(fs.heads), B_.raw(fs.ws), fs.ws.numel(), ctypes.byref(fs._key_base))
            return
        self.fs.sort_ids(uid, iid)


class _DimShardedStep:
    """Skeleton shared by the pairwise and the pointwise step: pack -> all-gather -> unpack -> partial scores -> all-reduce ->
    gradients + sort + row-wise applies.  ``third_is_label``: the third per-row array is an fp32 label, not an id."""
    third_is_label = False

    def _setup(self, user_cols, item_cols, batch_per_rank, group, ops, stream):
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.U, self.I = user_cols, item_cols
        self.Ds = user_cols.shape[1]
        Bg = int(batch_per_rank) * self.world
        self.max_batch = Bg
        dev = user_cols.device
        self.ops = ops
        # two slots of exchange buffers: ``prefetch`` fills the one the running step is not reading
        self._slots = [{'ids': torch.empty(3 * Bg, device=dev, dtype=torch.int64),
                        'labels': torch.empty(Bg, device=dev, dtype=torch.float32) if self.third_is_label else None,
                        'ids32': torch.empty(3 * int(batch_per_rank), device=dev, dtype=torch.int32),
                        'gath32': torch.empty(3 * Bg, device=dev, dtype=torch.int32)} for _ in range(2)]
        self._cur, self._pf, self._side, self._checked_size = 0, None, None, -1
        self.diff = torch.empty(Bg + 2, device=dev, dtype=torch.float32)
        self.out = self.ops.out
        self.stream = stream
        self._prof = None
        self.force_collectives = False           # run the (identity) collectives at world 1 too: exercises the RCCL calls on one GPU

    @property
    def ustate(self):
        return self.ops.fs.ustate

    @property
    def istate(self):
        return self.ops.fs.istate

    def loss_value(self):
        return self.out[0]

    def profile(self, on=True):
        self._prof = {'bytes': 0, 'events': []} if on else None

    def exchange_stats(self):
        """-> (bytes this rank sent to other ranks, milliseconds inside collectives) since ``profile()``; synchronises."""
        if not self._prof:
            return 0, 0.0
        torch.cuda.synchronize()
        return self._prof['bytes'], sum(a.elapsed_time(b) for a, b in self._prof['events'])

    def _on_stream(self):
        return torch.cuda.stream(self.stream) if self.stream is not None else contextlib.nullcontext()

    @contextlib.contextmanager
    def _timed(self, nbytes):
        if not self._prof or not self.U.is_cuda:
            yield
            return
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        yield
        b.record()
        self._prof['bytes'] += int(nbytes)
        self._prof['events'].append((a, b))

    @property
    def ids(self):
        """Field-major global ids of the step that ran last ([3 * B_global] int64; rows [B_global:] unused for pointwise rows)."""
        return self._slots[self._cur]['ids']

    def _exchange(self, slot, a, b, c):
        """pack -> all-gather -> unpack of one batch's rows into ``slot``; returns the global (a, b, c)."""
        G, ops, buf = self.world, self.ops, self._slots[slot]
        Bl = a.numel()
        Bg = G * Bl
        ops.pack_ids(a, b, c, buf['ids32'][:3 * Bl])
        with self._timed(3 * 4 * Bl * (G - 1)):
            dist.all_gather_into_tensor(buf['gath32'][:3 * Bg], buf['ids32'][:3 * Bl], group=self.group)
        idv = buf['ids'][:3 * Bg].view(3, Bg)                                  # field-major global rows of this step
        if self.third_is_label:
            ops.unpack_ids(buf['gath32'][:3 * Bg], G, Bl, idv, buf['labels'][:Bg])
            return idv[0], idv[1], buf['labels'][:Bg]
        ops.unpack_ids(buf['gath32'][:3 * Bg], G, Bl, idv)
        return idv[0], idv[1], idv[2]

    def prefetch(self, a, b, c):
        """Start the id exchange of the NEXT batch now, on a side stream, so that it runs under the current step's kernels (ids do
        not depend on the model).  ``step`` recognises the batch by identity of ``a`` and picks the gathered ids up."""
        if not (self.world > 1 or (self.force_collectives and dist.is_initialized())):
            return
        slot = 1 - self._cur
        if not self.U.is_cuda:                       # CPU (gloo tests): the same double-buffered hand-over, exchanged in place
            self._pf = (a, slot, self._exchange(slot, a, b, c), None)
            return
        if self._side is None:
            self._side = torch.cuda.Stream(device=self.U.device)
        self._side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(self._side):
            got = self._exchange(slot, a, b, c)
            ev = torch.cuda.Event()
            ev.record()
        self._pf = (a, slot, got, ev)

    def step(self, a, b, c, next_batch=None):
        """Pairwise: (uid, pid, nid); pointwise: (uid, iid, label).  The rank's own rows, the same count on every rank.
        ``next_batch``: the following step's (a, b, c), if known -- its id exchange is started first and overlaps this step."""
        G, grp, ops = self.world, self.group, self.ops
        Bl = a.numel()
        Bg = G * Bl
        assert Bg <= self.max_batch
        if self.stream is not None:
            self.stream.wait_stream(torch.cuda.current_stream())           # the ids were produced on the caller's stream
        comm = G > 1 or (self.force_collectives and dist.is_initialized())
        if comm and Bl != self._checked_size:
            # a rank with a different row count would make the fixed-size all-gather undefined: compare once per batch size
            sizes = torch.tensor([Bl], device=a.device, dtype=torch.int64)
            allsz = [torch.empty_like(sizes) for _ in range(G)]
            dist.all_gather(allsz, sizes, group=grp)
            got = [int(t.item()) for t in allsz]
            if any(x != Bl for x in got):
                raise ValueError(f'dimension-sharded step needs the same number of rows on every rank, got {got}')
            self._checked_size = Bl
        with self._on_stream():
            if comm:
                if self._pf is not None and self._pf[0] is a:
                    _a, slot, (a, b, c), ev = self._pf
                    if ev is not None:
                        torch.cuda.current_stream().wait_event(ev)
                    self._cur = slot
                else:
                    a, b, c = self._exchange(self._cur, a, b, c)
                self._pf = None
                if next_batch is not None:
                    self.prefetch(*next_batch)
            diff = self.diff[:Bg + 2]
            ops.partial(a, b, c, diff)
            work = None
            if comm:
                if self._prof:
                    self._prof['bytes'] += 2 * 4 * (Bg + 2) * (G - 1) // G          # ring all-reduce: reduce-scatter + all-gather
                work = dist.all_reduce(diff, group=grp, async_op=True)
            ops.presort(a, b, c)                      # the id sort needs no scores: it runs while the links carry them
            if work is not None:
                work.wait()                           # stream-level wait: the host does not block
            return ops.grad_apply(a, b, c, diff)


class DimShardedBPRStep(_DimShardedStep):
    """``user_cols`` / ``item_cols``: this rank's column slices [rows, D / world] (``dim_shard_of``).  ``step(uid, pid, nid)`` takes
    the rank's own batch (global row ids, the same length on every rank) and returns out (view; [0] = total loss of the GLOBAL
    batch, identical on every rank).  ``ops``: compute stand-in for the CPU (gloo) tests; default = the native kernels."""

    def __init__(self, user_cols, item_cols, batch_per_rank, opt='adam', lr=1e-3, betas=(0.9, 0.999), eps=1e-8,
                 weight_decay=0.0, gamma=1e-10, reg_weight=0.0, group=None, ops=None, stream=None, user_state=None,
                 item_state=None):
        world = dist.get_world_size(group) if dist.is_initialized() else 1
        if ops is None:
            ops = NativeDimOps(user_cols, item_cols, int(batch_per_rank) * world, opt=opt, lr=lr, betas=betas, eps=eps,
                               weight_decay=weight_decay, gamma=gamma, reg_weight=reg_weight, user_state=user_state,
                               item_state=item_state)
        self._setup(user_cols, item_cols, batch_per_rank, group, ops, stream)


class DimShardedPointStep(_DimShardedStep):
    """Pointwise counterpart (fused.FusedPointStep in the dimension layout): ``step(uid, iid, label)`` with recbole's pointwise
    rows -- MSE on the raw dot (EMCDR's MF latent factor model) or BCE on sigmoid(dot)."""
    third_is_label = True

    def __init__(self, user_cols, item_cols, batch_per_rank, loss='mse', opt='adam', lr=1e-3, betas=(0.9, 0.999), eps=1e-8,
                 weight_decay=0.0, reg_weight=0.0, group=None, ops=None, stream=None, user_state=None, item_state=None):
        world = dist.get_world_size(group) if dist.is_initialized() else 1
        if ops is None:
            ops = NativePointDimOps(user_cols, item_cols, int(batch_per_rank) * world, loss=loss, opt=opt, lr=lr, betas=betas,
                                    eps=eps, weight_decay=weight_decay, reg_weight=reg_weight, user_state=user_state,
                                    item_state=item_state)
        self._setup(user_cols, item_cols, batch_per_rank, group, ops, stream)
