"""How ``bench.py --gpus N`` lays BASELINE config C5 (EMCDR-BPR, two domains, row-wise Adam) over the ranks of a node: which
process groups are created, which step object every rank builds for which domain, and how one benchmark step drives them
(prefetched id exchange, two-domain pipelining).  Kept in the package -- not in bench.py -- so that the CPU (gloo) tests run this
exact sequence with stand-in arithmetic (tests/test_shard_gloo.py::test_bench_layouts_under_gloo) on every commit; the hardware
run is the first time RCCL carries more than one rank of it.

Layouts:
  'dim'  every rank holds D / G columns of every row; ids all-gathered (prefetched), one partial score per triple all-reduced.
         With ``domain_groups`` (even N >= 4) each domain gets one half of the ranks (G = N / 2, 2 B triples of its domain per rank
         and step); at N = 2 both domains are cut over both ranks (G = 2), so that the first multi-GPU point exercises the
         collectives instead of being two independent GPUs.
  'row'  rows r % N; user-aligned routing, row / gradient-row all-to-all, the two domains pipelined on their own streams / groups.
"""
import torch
import torch.distributed as dist


class C5Layout:
    def __init__(self):
        self.steps, self.tabs = {}, {}
        self.mode, self.my_dom, self.half, self.Ds = None, None, None, None
        self.pipeline = False
        self.prefetch = True

    # ---- one benchmark step = one batch of every domain this rank works on -------------------------------------------------
    def rank_domains(self):
        return (self.my_dom,) if self.mode == 'dim-groups' else ('source', 'target')

    def run(self, batches, i):
        """``batches``: list (pool) of {domain: (uid, pid, nid)}; step i uses batches[i % len]."""
        b = batches[i % len(batches)]
        if self.mode == 'dim-groups':
            st = self.steps[self.my_dom]
            if self.half > 1:                      # the next batch's id all-gather starts under this step's kernels
                st.step(*b[self.my_dom], next_batch=batches[(i + 1) % len(batches)][self.my_dom])
            else:
                st.step(*b[self.my_dom])
        elif self.mode == 'row':
            # the following batch's id-only stages run on a side stream behind this step's kernels (ShardedBPRStep.step_gen, direct form)
            nb = batches[(i + 1) % len(batches)] if self.prefetch else None
            kw = lambda d: {'next_batch': nb[d]} if (nb is not None and self.steps[d].direct) else {}  # noqa: E731
            if self.pipeline:
                from .shard import run_pipelined
                run_pipelined([self.steps[d].step_gen(*b[d], **kw(d)) for d in ('source', 'target')])
            else:
                for d in ('source', 'target'):
                    self.steps[d].step(*b[d], **kw(d))
        else:
            for d in ('source', 'target'):
                self.steps[d].step(*b[d])


def resolve(world, layout, D, domain_groups=True):
    """-> (mode, columns per rank or None).  mode in {'dim-groups', 'dim', 'row'}; falls back to 'row' when D does not cut into
    float4-wide column slices."""
    if layout == 'row':
        return 'row', None
    groups = domain_groups and world >= 4 and world % 2 == 0
    G = world // 2 if groups else world
    if D % (4 * G):
        return 'row', None
    return ('dim-groups' if groups else 'dim'), D // G


def make_groups(world, mode):
    """The process groups of layout ``mode`` ('dim-groups' | 'dim' | 'row').  ``dist.new_group`` must be called by EVERY rank in the same
    order, so a caller that may abandon a layout half-way (preflight.try_layouts) creates the groups of all its candidates up front,
    on every rank, and hands them to ``build`` -- a rank that failed early would otherwise leave the others' later ``new_group`` calls
    paired with the wrong partners.  (With RCCL a group costs nothing until its first collective.)"""
    if mode == 'dim-groups':
        half = world // 2
        return {'source': dist.new_group(list(range(half))), 'target': dist.new_group(list(range(half, world)))}
    g = {d: dist.new_group(list(range(world))) for d in ('source', 'target')}
    if mode == 'row':        # the id-only stages of the row layout run on a side stream: their own communicators (shard.ShardedBPRStep.index_group)
        g.update({d + '_ix': dist.new_group(list(range(world))) for d in ('source', 'target')})
    return g


def build(world, rank, layout, D, B, n_users, n_items, make_table, step_kw, domain_groups=True, pipeline=True, dedup=True, device=None,
          dim_ops=None, row_ops=None, plain_step=None, groups=None, row_comm=None):
    """``make_table(name, rows, cols, total_cols)`` -> this rank's fp32 table [rows, cols] (name in su, si, tu, ti).
    ``step_kw``: optimizer / loss keywords of the step classes (opt, reg_weight, lr ...).
    ``dim_ops(user_cols, item_cols, max_global_batch)`` / ``row_ops()``: compute stand-ins for the CPU tests (None: native kernels).
    ``plain_step(user_tab, item_tab, max_batch)``: the single-GPU step class for a one-rank domain group (default FusedBPRStep).
    ``row_comm(group)`` -> shard.CabiComm: the row layout's exchanges through the C ABI's communicator (None: torch.distributed)."""
    from .dimshard import DimShardedBPRStep
    from .shard import ShardedBPRStep, shard_rows
    lay = C5Layout()
    lay.mode, lay.Ds = resolve(world, layout, D, domain_groups)
    lay.pipeline = bool(pipeline)
    cuda = device is not None and torch.device(device).type == 'cuda'
    mk_stream = (lambda: torch.cuda.Stream(device=device)) if (cuda and pipeline) else (lambda: None)
    if lay.mode == 'dim-groups':
        half = lay.half = world // 2
        groups = groups if groups is not None else make_groups(world, 'dim-groups')
        lay.my_dom = 'source' if rank < half else 'target'
        k = lay.my_dom[0]
        lay.tabs = {k + 'u': make_table(k + 'u', n_users, lay.Ds, D), k + 'i': make_table(k + 'i', n_items, lay.Ds, D)}
        U, I = lay.tabs[k + 'u'], lay.tabs[k + 'i']
        ops = dim_ops(U, I, 2 * B * half) if dim_ops is not None else None
        lay.steps = {lay.my_dom: DimShardedBPRStep(U, I, 2 * B, group=groups[lay.my_dom], ops=ops, **step_kw)}
        lay.groups = groups
    elif lay.mode == 'dim':
        lay.tabs = {n: make_table(n, r, lay.Ds, D) for n, r in (('su', n_users), ('si', n_items), ('tu', n_users), ('ti', n_items))}
        # one process group (= one communicator) and one stream per domain: the two domain steps touch disjoint tables and have no
        # host sync inside, so they queue up side by side and one's collectives overlap the other's kernels
        lay.groups = groups if groups is not None else make_groups(world, 'dim')
        for d in ('source', 'target'):
            U, I = lay.tabs[d[0] + 'u'], lay.tabs[d[0] + 'i']
            ops = dim_ops(U, I, B * world) if dim_ops is not None else None
            lay.steps[d] = DimShardedBPRStep(U, I, B, group=lay.groups[d], ops=ops, stream=mk_stream(), **step_kw)
    else:
        lay.tabs = {n: make_table(n, shard_rows(r, world, rank), D, D) for n, r in (('su', n_users), ('si', n_items), ('tu', n_users), ('ti', n_items))}
        lay.groups = groups if groups is not None else make_groups(world, 'row')
        for d in ('source', 'target'):
            lay.steps[d] = ShardedBPRStep(lay.tabs[d[0] + 'u'], lay.tabs[d[0] + 'i'], n_users, n_items, B, group=lay.groups[d],
                                          ops=row_ops() if row_ops is not None else None, stream=mk_stream(), dedup=dedup,
                                          comm=row_comm(lay.groups[d]) if row_comm is not None else None, index_group=lay.groups.get(d + '_ix'), **step_kw)
    return lay
