#!/usr/bin/env python3
"""bench.py -- training interactions/sec (+ full-sort items scored/sec) of the MI355X cross-domain hot path.

Contract (driver): `python bench.py --gpus N --steps K --warmup W`; for N>1 launched under torch.distributed.run, one
rank per GPU over RCCL.  W untimed warm-up steps, then EXACTLY K steps between barrier+synchronize brackets, MAX over
ranks, rank 0 prints ONE JSON line.

Workload (default `c5`, BASELINE.json configs[4], the configuration north_star quotes its targets on; it fits one
MI355X): EMCDR-BPR, D=128, union id space of 50,000,001 users x 20,000,001 items (10 M items per domain), fp32
xavier-normal tables, synthetic uniform interaction streams (seed 2022).  One STEP = one SOURCE-domain batch + one
TARGET-domain batch of `--batch` (u, i+, i-) triples each: fused gather -> BPR+EmbLoss -> per-row gradients -> row-wise
Adam update of the touched rows (csrc/cdr_step.hip).  Inputs (tables, id batches) are resident in HBM before the timed
region.  With N>1 the tables are row-sharded (row r on rank r % N) and every rank contributes its own batch
(weak scaling in batch; table size fixed); rows/gradients travel by RCCL all-to-all (shard.py).

`--workload c1|c2|c3|c4` run BASELINE configs[0..3] (CMF and EMCDR at ml-1m->ml-100k sizes, CoNet Amazon sizes, BiTGCF Douban sizes)
through the drop-in class contract (autograd + dense Adam in torch.optim.Adam's semantics), the whole step replayed as one hipGraph.
"""
import argparse
import json
import os
import sys
import time
T_START = time.perf_counter()

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured float4 copy)
FP32_MFMA_PEAK_TFLOPS = 157.3



CONTRACT_KEYS = ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline',
                 'dtype', 'data')
DETAIL_FILE = 'bench_detail.json'
LINE_LIMIT = 4096


def _clip(s, n):
    s = str(s)
    return s if len(s) <= n else s[:n - 3] + '...'


def compact_line(result, detail_file=DETAIL_FILE, limit=LINE_LIMIT):
    """The ONE line the driver parses: the contract keys, `config`, `roofline` (with `traffic` as a number of bytes or null),
    `cpu_baseline`, the wall time of the invocation and the name of the file every leg went to.  Everything else bench.py
    measures (per-kernel brackets, the OVERLAP / k=4 / grid / full-sort legs, the C1-C4 legs, the end-to-end leg, both N>1
    layouts) is in `detail_file`.  Round 4's line carried all of that inline (24.7 KB) and the driver could not parse it
    (BENCH_r04.json `parsed: null`); this one is bounded by `limit` bytes and a test holds it there."""
    line = {k: result.get(k) for k in CONTRACT_KEYS}
    cfg = result.get('config') or {}
    line['config'] = {k: (_clip(v, 320) if isinstance(v, str) else v) for k, v in cfg.items()
                      if isinstance(v, (str, int, float, bool)) or v is None}
    rf = result.get('roofline')
    if isinstance(rf, dict):
        tr = rf.get('traffic')
        out = {k: rf.get(k) for k in ('bound', 'kernel', 'achieved', 'peak', 'unit', 'frac', 'avg_launch_ms', 'algorithmic_bytes') if k in rf}
        out['traffic'] = tr.get('bytes') if isinstance(tr, dict) else tr
        if isinstance(tr, dict) and tr.get('source'):
            out['traffic_source'] = _clip(tr['source'], 200)
        if out.get('traffic') and rf.get('algorithmic_bytes'):
            out['traffic_over_algorithmic'] = round(out['traffic'] / rf['algorithmic_bytes'], 4)
        line['roofline'] = out
    cb = result.get('cpu_baseline')
    if isinstance(cb, dict):
        line['cpu_baseline'] = {k: (_clip(v, 260) if isinstance(v, str) else v) for k, v in cb.items()
                                if k in ('value', 'unit', 'cores', 'kind', 'sample', 'host_cores', 'cpu_model')}
        rfm = cb.get('reference_formulation')
        if isinstance(rfm, dict) and rfm.get('value'):
            # the reference's OWN formulation of the step (dense autograd gradients + torch.optim.Adam over every row), same host, same shape
            line['cpu_baseline']['reference_formulation'] = {'value': rfm['value'], 'unit': rfm.get('unit'), 'cores': rfm.get('cores')}
        if cb.get('spread') is not None:
            line['cpu_baseline']['spread'] = round(cb['spread'], 3)
        if cb.get('value') and result.get('value'):
            line['vs_cpu'] = round(result['value'] / cb['value'], 1)
    # a handful of scalars a reader wants beside the headline without opening the detail file
    extra = {}
    rs = result.get('roofline_step')
    if isinstance(rs, dict) and rs.get('frac') is not None:
        extra['step_frac_of_hbm_peak'] = rs['frac']
    fs = result.get('fullsort')
    if isinstance(fs, dict):
        for name, leg in fs.items():
            if isinstance(leg, dict) and isinstance(leg.get('items_per_s'), (int, float)):
                extra['fullsort_items_per_s_' + name.replace('=', '')] = leg['items_per_s']
    mfp = result.get('mf_pointwise')
    if isinstance(mfp, dict) and mfp.get('rows_per_s'):
        extra['mf_pointwise_rows_per_s'] = mfp['rows_per_s']
    ing = result.get('ingest')
    if isinstance(ing, dict) and ing.get('value'):
        extra['ingest_tokens_per_s'] = ing['value']
    if result.get('north_star_layout'):
        extra['north_star_layout'] = result['north_star_layout']
    lay = result.get('layouts')
    if isinstance(lay, dict):
        extra['layouts'] = {k: (None if v is None else {'value': v.get('value'), 'ms_per_step': v.get('ms_per_step'),
                                                        'sharding': _clip(v.get('sharding', ''), 60)}) for k, v in lay.items()}
    lf = result.get('layout_fallback')
    if isinstance(lf, dict):
        extra['layout_fallback'] = {k: lf.get(k) for k in ('used', 'fell_back', 'ranks_seen', 'comm') if k in lf}
    ex = result.get('exchange')
    if isinstance(ex, dict):
        extra['exchange'] = {k: v for k, v in ex.items() if isinstance(v, (int, float))}
    legs = result.get('configs')
    if isinstance(legs, dict):
        extra['config_legs_ms_per_step'] = {k: round(v['ms_per_step'], 5) for k, v in legs.items() if isinstance(v, dict) and v.get('ms_per_step')}
    if result.get('leg_errors'):
        extra['leg_errors'] = sorted(result['leg_errors'])
    line.update(extra)
    for k in ('deterministic_backward', 'nondeterministic_legs', 'bench_wall_s'):
        if k in result:
            line[k] = result[k]
    line['detail_file'] = detail_file
    # bounded whatever the legs put in: drop the optional scalars first, then clip the two prose fields harder
    for drop in (tuple(extra), ('traffic_source',), ()):
        txt = json.dumps(line, separators=(',', ':'))
        if len(txt) <= limit:
            return txt
        for k in drop:
            line.pop(k, None)
            (line.get('roofline') or {}).pop(k, None)
    line['config'] = {k: (_clip(v, 120) if isinstance(v, str) else v) for k, v in line['config'].items()}
    if 'cpu_baseline' in line:
        line['cpu_baseline']['sample'] = _clip(line['cpu_baseline'].get('sample', ''), 120)
    txt = json.dumps(line, separators=(',', ':'))
    assert len(txt) <= limit, len(txt)
    return txt


def emit(result, real_stdout=None, detail_dir=None, detail_file=None):
    """Write every leg to bench_detail.json (repo root; also gpurun_out/ when that directory exists, so a gpurun call brings it
    back; `--detail-file PATH` names another place) and print the compact line as the LAST line of stdout."""
    if detail_file:
        paths = [os.path.abspath(detail_file)]
    else:
        paths = [os.path.join(detail_dir or ROOT, DETAIL_FILE)]
        go = os.path.join(ROOT, 'gpurun_out')
        if detail_dir is None and os.path.isdir(go):
            paths.append(os.path.join(go, DETAIL_FILE))
    for p in paths:
        try:
            with open(p, 'w') as f:
                json.dump(result, f, indent=1)
                f.write('\n')
        except OSError as e:
            print('bench: could not write %s: %r' % (p, e), file=sys.stderr)
    txt = compact_line(result, detail_file=detail_file or DETAIL_FILE) + '\n'
    if real_stdout is not None:
        os.write(real_stdout, txt.encode())
    else:
        sys.stdout.write(txt)
        sys.stdout.flush()


def exact_sum(t, step=1 << 28):
    """fp64 sum of a (large) fp32 tensor in slices: ``torch.sum(t, dtype=float64)`` materialises an fp64 copy of the whole operand on this
    build (48 GiB for a 25.6 GB table), which does not fit beside the C5 tables once a few legs have run.  Slice order is fixed: the digits
    are comparable between invocations of the same bench.py."""
    import torch
    flat = t.detach().reshape(-1)
    total = torch.zeros((), device=flat.device, dtype=torch.float64)
    for i in range(0, flat.numel(), step):
        total += torch.sum(flat[i:i + step], dtype=torch.float64)
    return float(total)

def host_cores():
    """Cores this process may actually use: the scheduler affinity mask, capped by the cgroup CPU quota (a container that sees 256
    CPUs but owns 8 of them oversubscribes 32x with torch.set_num_threads(os.cpu_count()))."""
    try:
        n = len(os.sched_getaffinity(0))
    except Exception:
        n = os.cpu_count() or 1
    for path in ('/sys/fs/cgroup/cpu.max', '/sys/fs/cgroup/cpu/cpu.cfs_quota_us'):
        try:
            with open(path) as f:
                txt = f.read().split()
            if path.endswith('cpu.max'):
                if txt[0] != 'max':
                    n = min(n, max(1, -(-int(txt[0]) // int(txt[1]))))
            else:
                q = int(txt[0])
                if q > 0:
                    with open('/sys/fs/cgroup/cpu/cpu.cfs_period_us') as f2:
                        n = min(n, max(1, -(-q // int(f2.read().split()[0]))))
            break
        except Exception:
            continue
    return max(1, n)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=30)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--workload', default='c5', choices=['c5', 'c1', 'c2', 'c3', 'c4'])
    ap.add_argument('--dense-adam', action='store_true', help='c3: the literal O(table) dense Adam sweep instead of its deferred row-wise form')
    ap.add_argument('--batch', type=int, default=1 << 20, help='triples per domain per rank per step (c5)')
    ap.add_argument('--opt', default='adam', choices=['adam', 'sgd'])
    ap.add_argument('--users', type=int, default=50_000_001)
    ap.add_argument('--items-per-domain', type=int, default=10_000_000)
    ap.add_argument('--dim', type=int, default=128)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--detail-file', default=None, help='where every leg of the run is written as JSON (default: bench_detail.json at the repo root)')
    ap.add_argument('--no-fullsort', action='store_true')
    ap.add_argument('--headline-only', action='store_true', help='c5, N=1: the headline step alone (no OVERLAP / per-positive / grid / full-sort / C1-C4 legs, no CPU baseline): the command whose rocprofv3 --stats averages are the headline kernels\' own (profiles/*_headline_kernel_stats.csv)')
    ap.add_argument('--no-config-legs', action='store_true', help='c5, N=1: skip the compact C1-C4 legs (BASELINE configs[0..3]) behind the headline')
    ap.add_argument('--no-e2e', action='store_true', help='c5, N=1: skip the end-to-end leg (CrossDomainTrainer.fit over SOURCE / TARGET / OVERLAP epochs + evaluate at the headline table sizes)')
    ap.add_argument('--only-e2e', action='store_true', help='c5, N=1: the end-to-end leg alone')
    ap.add_argument('--no-map', action='store_true', help='c5: skip the OVERLAP-phase leg (profiling runs of the BPR step alone)')
    ap.add_argument('--no-ingest', action='store_true', help='c5, N=1: skip the ingest leg (overlap id remap of the C5 id space on the device)')
    ap.add_argument('--only-ingest', action='store_true', help='c5, N=1: the ingest leg alone')
    ap.add_argument('--cpu-seconds', type=float, default=15.0)
    ap.add_argument('--no-graph', action='store_true', help='c3/c4: run the step eagerly instead of replaying a hipGraph')
    ap.add_argument('--full-last-layer', action='store_true', help='c4: evaluate every row of the last propagation layer (the reference\'s order) instead of the rows the loss gathers')
    ap.add_argument('--no-pipeline', action='store_true', default=bool(int(os.environ.get('CDR_NO_PIPELINE', '0'))),
                    help='sharded path: run the two domain steps back to back on one stream')
    ap.add_argument('--force-shard', action='store_true', help='run the sharded exchange path even with 1 rank')
    ap.add_argument('--no-prefetch', action='store_true', help="row shard: do not run the next batch's id-only stages on a side stream behind this step's kernels (A/B)")
    ap.add_argument('--no-direct', action='store_true', help="row shard: round 5's staged form (A/B)")
    ap.add_argument('--routed-batch-loss', action='store_true', help='c4 row shard: the routed batch loss even with one rank (--force-shard A/B)')
    ap.add_argument('--replicated-batch-loss', action='store_true', help='c4 row shard: every rank scores the whole batch on the all-gathered stacked tables '
                                                                         '(round-3 form) instead of its B/N slice on routed rows')
    ap.add_argument('--shard', default='row', choices=['dim', 'row'],
                    help='N>1 layout of the C5 tables whose numbers the line carries (the other one is timed into `layouts`): row = rows r %% N with the '
                         'row / gradient-row all-to-all exchange (BASELINE configs[4] / north_star; the default since round 6); dim = every rank '
                         'holds D/N columns of every row (ids all-gathered, one partial score per triple all-reduced)')
    ap.add_argument('--no-domain-groups', action='store_true',
                    help='--shard dim: shard BOTH domains over all N ranks (D/N columns each) instead of giving each domain one half of '
                         'the ranks (D/(N/2) columns, twice the batch per rank)')
    ap.add_argument('--replica-dp', action='store_true', help='c4 with N>1: replica data parallelism with sharded Adam (dp.py) instead of the row-sharded graph')
    ap.add_argument('--single-stream', action='store_true', help='c5, N=1: enqueue the TARGET domain step behind the SOURCE domain step on one stream (default: two streams)')
    ap.add_argument('--comm', default='torch', choices=['torch', 'cabi'],
                    help='c5 row shard: the data-path exchanges through torch.distributed (RCCL) or through the C ABI\'s own communicator '
                         '(cdr_comm_init + cdr_a2a_ids / cdr_a2a_rows / cdr_allreduce_sum_f32; falls back to torch if it does not come up)')
    ap.add_argument('--single-layout', action='store_true', help='N>1: time only the --shard layout (default: both, in one record)')
    ap.add_argument('--preflight-seconds', type=float, default=45.0, help='N>1: deadline for a candidate layout to create its groups and run its first two steps on every rank')
    ap.add_argument('--no-layout-fallback', action='store_true', help='N>1: fail instead of trying the next layout / independent replicas')
    ap.add_argument('--no-dedup', action='store_true', help='sharded path: exchange one row per occurrence instead of one per distinct item')
    return ap.parse_args()


def dist_setup(args):
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if args.gpus > 1 and world == 1:
        raise SystemExit('--gpus N>1 must be launched with torch.distributed.run (one rank per GPU)')
    # CDR_BENCH_SHARED_GPU=1: every rank on cuda:0 over gloo -- a functional check of the N>1 code path on a 1-GPU box
    # (RCCL refuses two ranks on one device); never a performance number, and the JSON line says so.
    shared = bool(int(os.environ.get('CDR_BENCH_SHARED_GPU', '0')))
    if shared:
        local = 0
        if world > 1 and args.workload == 'c5':
            # every rank allocates its own tables on the ONE device: refuse sizes that cannot fit `world` times (a run that drives the box out of
            # VRAM takes the box down with it) -- the functional checks pass small --users / --items-per-domain / --batch
            per_rank = (args.users + 2 * args.items_per_domain) * args.dim * 4 * 3 / max(world // 2, 1)       # tables + two Adam moments, upper bound
            total = torch.cuda.get_device_properties(0).total_memory
            if per_rank * world > 0.6 * total:
                raise SystemExit('CDR_BENCH_SHARED_GPU=1 with %d ranks needs ~%.0f GB on one %.0f GB device at these table sizes: pass smaller '
                                 '--users / --items-per-domain (this mode is a functional check, never a measurement)'
                                 % (world, per_rank * world / 1e9, total / 1e9))
    torch.cuda.set_device(local)
    if world > 1 or args.force_shard:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_PORT', '29533')
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        import datetime
        if shared:
            dist.init_process_group('gloo', rank=rank, world_size=world, timeout=datetime.timedelta(minutes=5))
        else:
            # no device_id: communicators are then created lazily with ncclCommInitRank, per group, by its members only --
            # the long-standing path; binding a device would create every sub-group with ncclCommSplit instead
            dist.init_process_group('nccl', rank=rank, world_size=world,
                                    timeout=datetime.timedelta(minutes=5))  # a wedged collective aborts instead of hanging
        global CTRL
        from recbole_cdr_amd import preflight
        CTRL = preflight.control_group()
    return world, rank, local


CTRL = None            # gloo control group (preflight.control_group): barriers, verdicts and timing scalars never depend on RCCL


def barrier(world):
    torch.cuda.synchronize()
    if world > 1:
        import torch.distributed as dist
        dist.barrier(group=CTRL)
    torch.cuda.synchronize()


def ctrl_max(t):
    """MAX over ranks of a small tensor of timing scalars, over the control group (host copies); returned on t's device."""
    import torch.distributed as dist
    c = t.detach().to('cpu', copy=True)
    dist.all_reduce(c, op=dist.ReduceOp.MAX, group=CTRL)
    return c.to(t.device)


class LayoutUnavailable(RuntimeError):
    """No candidate layout came up on every rank within the preflight deadline; ``attempts`` says what failed where."""

    def __init__(self, attempts):
        super().__init__('no multi-GPU layout came up: %s' % [(a['layout'], a['errors']) for a in attempts])
        self.attempts = attempts


def xavier_table(rows, D, total_rows, gen, dev):
    # xavier_normal_ on the [total_rows, D] table: std = sqrt(2 / (rows + D))  (recbole xavier_normal_initialization)
    std = (2.0 / (total_rows + D)) ** 0.5
    t = torch.empty(rows, D, device=dev, dtype=torch.float32)
    t.normal_(0.0, std, generator=gen)
    return t


# ------------------------------------------------------------------------------------------------------ C5 workload
def run_c5(args, world, rank, dev):
    import torch.distributed as dist
    from recbole_cdr_amd.fused import FusedBPRStep
    from recbole_cdr_amd import functional as F_
    D, B = args.dim, args.batch
    OU, TOI = args.users, args.items_per_domain
    n_users, n_items = OU, 1 + 2 * TOI                     # union sizes: OI = 1 (PAD), TOI = SOI = 10 M
    # memory guard: tables (x3 with Adam moments) must fit what this GPU has free; shrink the USER count if they do not
    # (never silently: the workload string below always states the sizes actually used)
    free_b, _total_b = torch.cuda.mem_get_info(dev)
    state_mult = 3 if args.opt == 'adam' else 1
    need = lambda nu: 2 * 4.0 * D * (nu + n_items) * state_mult / max(world, 1) + 8e9
    while need(n_users) > free_b and n_users > 2_000_000:
        n_users = n_users // 2 + 1
    OU = n_users
    gen = torch.Generator(device=dev); gen.manual_seed(2022 + rank)
    sharded = world > 1 or args.force_shard
    lay = None
    attempts = None
    pool = 4

    def make_batches(dom_groups_, my_dom_):
        # synthetic interaction streams: users ~ U{1..OU-1}; target items [1, TOI], source items [TOI+1, 2 TOI]
        out = []
        for _ in range(pool):
            b = {}
            for dom, lo in (('source', 1 + TOI), ('target', 1)):
                if dom_groups_ and dom != my_dom_:
                    continue
                nb = 2 * B if dom_groups_ else B
                u = torch.randint(1, OU, (nb,), device=dev, generator=gen)
                p = torch.randint(lo, lo + TOI, (nb,), device=dev, generator=gen)
                n = torch.randint(lo, lo + TOI, (nb,), device=dev, generator=gen)
                b[dom] = (u, p, n)
            out.append(b)
        return out
    if sharded:
        # which groups / step objects / streams every rank builds, and how a step drives them, lives in the package
        # (recbole_cdr_amd/c5_layouts.py) so that the CPU gloo tests run this exact sequence with stand-in arithmetic.  The layout
        # is brought up under a watchdog (recbole_cdr_amd/preflight.py): group creation + the first two steps of a candidate within
        # --preflight-seconds on EVERY rank, else the next candidate: dim-groups -> dim -> row; none: LayoutUnavailable (main() then
        # runs N independent replicas and still prints the line).
        from recbole_cdr_amd import c5_layouts, preflight
        first, _ = c5_layouts.resolve(world, args.shard, D, not args.no_domain_groups)
        if args.shard == 'dim' and first == 'row':
            print('bench: --dim %d does not cut into float4 slices over %d ranks; using --shard row' % (D, world), file=sys.stderr)
        if args.comm == 'cabi':
            args.shard = 'row'                             # the C ABI's exchanges are the row shard's
            first = 'row'
        # fallback order behind the first candidate: the row shard first (north_star's layout) then the dimension layouts, or -- with
        # --shard dim -- the dimension layouts then the row shard; a candidate that does not resolve for this world / D is left out
        dims = [m for m in ('dim-groups', 'dim') if c5_layouts.resolve(world, 'dim', D, m == 'dim-groups')[0] == m]
        if first == 'row':
            chain = ['row'] + ([] if args.shard == 'dim' else dims)       # (--shard dim that resolved to row: D does not cut -- nothing behind it)
        else:
            chain = dims[dims.index(first):] + ['row']
        if args.comm == 'cabi':
            chain = ['row-cabi', 'row']
        if args.single_layout or args.no_layout_fallback:
            chain = chain[:1]

        def make_table(name, rows, cols, total_cols):
            total_rows = n_users if name[1] == 'u' else n_items
            return torch.empty(rows, cols, device=dev, dtype=torch.float32).normal_(0.0, (2.0 / (total_rows + total_cols)) ** 0.5, generator=gen)

        # every candidate's process groups, created up front by every rank in the same order (c5_layouts.make_groups says why)
        all_groups = {name: c5_layouts.make_groups(world, name) for name in chain}

        def build(name):
            row_comm = None
            if name == 'row-cabi':
                from recbole_cdr_amd.shard import CabiComm
                row_comm = lambda g: CabiComm(g, dev)  # noqa: E731
            cand = c5_layouts.build(world, rank, 'row' if name.startswith('row') else 'dim', D, B, n_users, n_items, make_table,
                                    dict(opt=args.opt, reg_weight=0.01), domain_groups=name == 'dim-groups', pipeline=not args.no_pipeline,
                                    dedup=not args.no_dedup, device=dev, groups=all_groups[name], row_comm=row_comm)
            cand.batches = make_batches(cand.mode == 'dim-groups', cand.my_dom)
            cand.prefetch = not args.no_prefetch
            if args.no_direct and cand.mode == 'row':
                for st in cand.steps.values():
                    st.direct = False
            return cand

        def first_steps(cand):
            # communicators are created lazily by their first collective (seconds, once): two set-up steps take that out of the way
            for i in range(2):
                cand.run(cand.batches, i)
            torch.cuda.synchronize()

        def cleanup(cand):
            cand.steps.clear(); cand.tabs.clear(); cand.batches = None
        if world > 1:
            name, lay, attempts = preflight.try_layouts(chain, build, first_steps, CTRL, seconds=args.preflight_seconds, device=dev, cleanup=cleanup)
            if name is None:
                raise LayoutUnavailable(attempts)
        else:
            lay = build(chain[0]); first_steps(lay)
        steps, tabs, batches = lay.steps, lay.tabs, lay.batches
        args.shard = 'row' if lay.mode == 'row' else 'dim'
    dim_mode = sharded and lay.mode in ('dim', 'dim-groups')
    dom_groups = sharded and lay.mode == 'dim-groups'
    if dom_groups:
        half, my_dom, Ds = lay.half, lay.my_dom, lay.Ds
    elif dim_mode:
        Ds = lay.Ds
    if not sharded:
        tabs = {k: xavier_table(r, D, r, gen, dev) for k, r in
                (('su', n_users), ('si', n_items), ('tu', n_users), ('ti', n_items))}
        steps = {'source': FusedBPRStep(tabs['su'], tabs['si'], B, opt=args.opt, reg_weight=0.01),
                 'target': FusedBPRStep(tabs['tu'], tabs['ti'], B, opt=args.opt, reg_weight=0.01)}
        batches = make_batches(False, None)

    from recbole_cdr_amd import binding as B_

    # N = 1: the SOURCE and the TARGET domain step of a benchmark step are enqueued on two HIP streams -- what the product's trainer does
    # with config['parallel_domains'] (trainer.py::_fit_domains_on_two_streams): disjoint tables and optimizer state, bit-identical to
    # one after the other; tails and small launches of one domain run under the other's kernels.  The roofline needs kernels timed ALONE:
    # their HIP-event brackets come from the same number of single-stream steps right behind the timed region (`single_stream`), and
    # `--headline-only` (the command the rocprofv3 summary is taken from) runs single-stream throughout, so the two agree.
    two_streams = not sharded and not args.single_stream and not args.headline_only
    if two_streams:
        dstreams = {d: torch.cuda.Stream(device=dev) for d in ('source', 'target')}
        for d in dstreams:
            with torch.cuda.stream(dstreams[d]):
                B_.ctx(dev)                                # the native context of each stream, created outside the timed region

    def one_step(i, serial=False):
        if lay is not None:
            lay.run(batches, i)
            return
        b = batches[i % pool]
        if two_streams and not serial:
            cur = torch.cuda.current_stream()
            for d in ('source', 'target'):
                dstreams[d].wait_stream(cur)
                with torch.cuda.stream(dstreams[d]):
                    steps[d].step(*b[d])
            for d in ('source', 'target'):
                cur.wait_stream(dstreams[d])
            return
        for dom in ('source', 'target'):
            steps[dom].step(*b[dom])

    for i in range(args.warmup):
        one_step(i)
    if sharded:
        for st in steps.values():
            st.profile(True)
    # HIP-event brackets around each hot kernel, recorded by the library on the stream the kernel is launched on
    if not two_streams:
        B_.timing_enable(dev, args.steps * 2 * 5 + 16)
    barrier(world)
    t0 = time.perf_counter()
    for i in range(args.steps):
        one_step(i)
    barrier(world)
    dt = time.perf_counter() - t0
    single = None
    if two_streams:
        for i in range(2):
            one_step(i, serial=True)
        B_.timing_enable(dev, args.steps * 2 * 5 + 16)
        torch.cuda.synchronize(); t1 = time.perf_counter()
        for i in range(args.steps):
            one_step(i, serial=True)
        torch.cuda.synchronize()
        single = (time.perf_counter() - t1) / args.steps
    timings = {}
    collected = B_.timing_collect(dev)
    if dim_mode and not dom_groups and not args.no_pipeline:
        # the two domain streams ran side by side in the timed region, so its per-kernel event times include sharing the
        # GPU with the other domain's kernels; the roofline figures come from two extra steps with the domains serialised
        prof = {d: st._prof for d, st in steps.items()}
        for st in steps.values():
            st._prof = None                         # not part of the timed region's exchange statistics
        for i in range(2):
            for dom in ('source', 'target'):
                steps[dom].step(*batches[i % pool][dom])
                torch.cuda.synchronize()
        for d, st in steps.items():
            st._prof = prof[d]
        collected = B_.timing_collect(dev)
    for name, ms in collected:
        timings.setdefault(name, []).append(ms)
    B_.timing_enable(dev, 0)
    mean_ms = lambda k: (sum(timings[k]) / len(timings[k])) if timings.get(k) else 0.0
    loss = float(next(iter(steps.values())).loss_value()) if dom_groups else float(steps['target'].loss_value()) if sharded else float(steps['target'].out6[0].item())

    tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
    if world > 1:
        tmax = ctrl_max(tmax)
    dt = float(tmax.item())
    interactions = 2 * B * args.steps * world
    result = {
        'metric': 'training interactions/sec', 'value': interactions / dt, 'unit': 'interactions/s',
        'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': dt / args.steps * 1e3,
        'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
        'config': {'workload': 'C5: EMCDR-BPR D=%d, %d users x %d items/domain (union tables %.1f GB fp32), '
                               'step = source batch + target batch of %d triples each per rank, fwd+bwd+row-wise %s'
                               % (D, OU - 1, TOI, 4.0 * D * 2 * (n_users + n_items) / 1e9, B, args.opt),
                   'batch_per_domain_per_rank': B, 'k_neg': 1, 'optimizer': 'rowwise-' + args.opt,
                   'sharding': 'none' if not sharded else
                   ('dimension x domain: ranks [0,%d) hold the source tables, ranks [%d,%d) the target tables, %d of %d columns of every row '
                    'per rank, 2 x %d triples of its domain per rank and step; ids all-gathered (prefetched), one partial score per triple '
                    'all-reduced inside the domain group' % (world // 2, world // 2, world, D // (world // 2), D, B)) if dom_groups else
                   ('dimension: %d of %d columns of every row per rank; ids all-gathered, one partial score per triple all-reduced, '
                    '2 domains on their own streams' % (D // world, D)) if dim_mode else
                   'row %% %d, user-aligned all-to-all%s, 2-domain pipelined' % (world, '' if args.no_dedup else ' of de-duplicated item rows')},
        'final_loss': loss,
    }
    if world == 1 and not sharded:
        # the trained tables as exact fp64 sums (after the timed region and its single-stream repeat, before any other leg trains on them):
        # two invocations with the same flags must print the same digits
        result['state_checksum'] = {k_: repr(exact_sum(v_)) for k_, v_ in tabs.items()}
    if single is not None:
        result['config']['streams'] = 'the two domain steps of a step on two HIP streams (CrossDomainTrainer parallel_domains); kernel brackets / roofline from single-stream steps'
        result['single_stream'] = {'ms_per_step': single * 1e3, 'value': 2 * B / single, 'unit': 'interactions/s', 'steps': args.steps,
                                   'what': 'the same steps with the TARGET domain step enqueued behind the SOURCE domain step on one stream, right after the '
                                           'timed region: the per-kernel HIP-event brackets (kernels, roofline) are taken here, with every kernel running alone'}
    if attempts is not None:
        # what the preflight saw: which candidate layouts were tried, how long each took to come up (group creation + two steps) and
        # every rank's error string for those that did not; ranks_seen = the size of each data group as RCCL itself counts it
        result['layout_fallback'] = {'used': lay.mode, 'attempts': attempts, 'fell_back': len(attempts) > 1}
        try:
            from recbole_cdr_amd import preflight
            result['layout_fallback']['ranks_seen'] = {d: preflight.ranks_seen(g, dev) for d, g in getattr(lay, 'groups', {}).items()
                                                       if d in lay.rank_domains()}
        except Exception as e:  # noqa: BLE001
            result['layout_fallback']['ranks_seen_error'] = repr(e)[:200]

    comms = {d: st.comm for d, st in steps.items() if getattr(st, 'comm', None) is not None} if sharded else {}
    if comms:
        # the row shard's data path went through the C ABI's communicator: its own rank count (cdr_comm_info) and how many calls it served
        lf = result.setdefault('layout_fallback', {'used': lay.mode, 'attempts': [], 'fell_back': False})
        lf['ranks_seen'] = {d: c.info()[1] for d, c in comms.items()}
        lf['comm'] = 'cabi'
        result['config']['comm'] = 'C ABI communicator (cdr_comm_init, cdr_a2a_ids / cdr_a2a_rows / cdr_allreduce_sum_f32); bucket counts over torch.distributed'
        result['cabi_calls_total'] = {k: sum(c.calls[k] for c in comms.values()) for k in ('cdr_a2a_ids', 'cdr_a2a_rows', 'cdr_allreduce_sum_f32')}
    elif sharded and args.comm == 'cabi':
        result.setdefault('layout_fallback', {})['comm'] = 'torch (the C ABI communicator did not come up: see attempts)'
    if sharded:
        # what the links carried and how long this rank's two domain streams spent inside all-to-alls (they overlap each other
        # and the other domain's kernels, so this is NOT additive with the kernel times: it says which side bounds the step)
        xb, xms = 0, 0.0
        for st in steps.values():
            b_, m_ = st.exchange_stats()
            xb += b_; xms += m_
        xt = torch.tensor([xb / args.steps, xms / args.steps], device=dev, dtype=torch.float64)
        if world > 1:
            xt = ctrl_max(xt)
        result['exchange'] = {'bytes_to_other_ranks_per_step_per_rank': float(xt[0]), 'collective_ms_per_step_max_rank': float(xt[1]),
                              'note': 'bytes: everything the data path sends; ms: HIP-event time of the collectives that are bracketed (dim: the id '
                                      'all-gather, which runs prefetched on a side stream -- the score all-reduce is asynchronous under the id sort and '
                                      'not bracketed; row: the 4 all-to-alls per domain, both domain streams summed)'}
    if rank == 0 and dim_mode:
        # every rank walks its group's GLOBAL batch on [rows, Ds] tables: the kernels are the single-GPU ones at width Ds
        st0 = steps['source']
        gsz = (world // 2) if dom_groups else world
        Bg = (2 * B if dom_groups else B) * gsz
        pn = st0.ids[Bg:3 * Bg] if gsz > 1 else torch.cat(batches[(args.steps - 1) % pool]['source'][1:])
        uniq_i = int(torch.unique(pn).numel())
        nmom = 6 if args.opt == 'adam' else 2
        if timings.get('bpr_fwd_apply_kernel'):
            # round 5: the second half of the dimension-sharded step is the one-GPU forward-and-update pass on Ds-column rows (fed with the
            # all-reduced scores): rows occurring once in the group's global batch are read-modify-written with their moments, the others
            # get one gradient row per occurrence (the headline's formula at width Ds)
            uu = st0.ids[:Bg] if gsz > 1 else batches[(args.steps - 1) % pool]['source'][0]
            cu = torch.unique(uu, return_counts=True)[1]
            ci = torch.unique(pn, return_counts=True)[1]
            su_, si_ = int((cu == 1).sum()), int((ci == 1).sum())
            row_b = 4 * Ds
            kname = 'bpr_fwd_apply_kernel'
            ms = mean_ms(kname)
            byts = (su_ + si_) * nmom * row_b + (3 * Bg - su_ - si_) * row_b
            what = 'bpr_fwd_apply_kernel (rank 0: the %d triples of its group\'s global batch on %d-column rows; %d user and %d item rows occur once)' % (Bg, Ds, su_, si_)
        else:
            kname = 'rowwise_apply_kernel(items)'
            ms = mean_ms(kname)
            byts = 2 * Bg * (8 + 4 * Ds) + uniq_i * nmom * 4 * Ds
            what = 'rowwise_apply_kernel(items) (rank 0: %d occurrences of its group\'s global batch on %d-column rows)' % (2 * Bg, Ds)
        gbs = byts / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
        result['roofline'] = {'bound': 'hbm', 'kernel': what, 'achieved': gbs, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': gbs / HBM_PEAK_GBS,
                              'avg_launch_ms': ms, 'algorithmic_bytes': byts, 'traffic': None}
        result['kernels'] = [{'kernel': k, 'avg_ms': mean_ms(k)} for k in
                             ('bpr_fwd_apply_kernel', 'batch_norms_kernel', 'occ_flags_kernel', 'bpr_fwd_grad_kernel', 'bpr_partial_diff_kernel',
                              'bpr_grad_from_diff_kernel', 'sort_ids', 'rowwise_apply_kernel(users)', 'rowwise_apply_kernel(items)') if timings.get(k)]
        if int(os.environ.get('CDR_BENCH_SHARED_GPU', '0')):
            result['data'] = 'synthetic; FUNCTIONAL CHECK ONLY: all ranks share cuda:0 over gloo'
    if rank == 0 and sharded and not dim_mode:
        # N > 1: the exchange adds all-to-alls between the kernels; the kernels themselves are the 1-GPU ones.  Buckets
        # are balanced in expectation (uniform ids), so one fwd_grad launch sees ~B triples: 3 rows read, GU + 2 GI rows
        # written (scatter mode).  Rank 0's own HIP-event durations.
        if timings.get('bpr_fwd_apply_kernel'):
            # round 5: the requester's half runs the one-GPU step's forward-and-update pass on the rows it holds (user rows of its own shard,
            # the item rows it was sent): per triple three rows read; user rows occurring once (most, at uniform ids) read-modify-written
            # with their Adam moments, the positive-side gradient row written for the segmented sum behind it
            kname = 'bpr_fwd_apply_kernel'
            ms = mean_ms(kname)
            nmom = 3 if args.opt == 'adam' else 1
            byts = B * (3 * 4 * D + 24) + B * (2 * nmom - 1) * 4 * D + B * 4 * D        # (as if every user row occurred once: an upper bound on the bytes)
            what = 'bpr_fwd_apply_kernel (rank 0: ~B triples per launch on the rows held after user-aligned routing; user rows updated in place)'
        else:
            kname = 'bpr_fwd_grad_kernel'
            ms = mean_ms(kname)
            byts = B * (3 * 4 * D + 24) + B * 3 * 4 * D
            what = 'bpr_fwd_grad_kernel (rank 0, scatter mode, ~B triples per launch)'
        gbs = byts / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
        result['roofline'] = {'bound': 'hbm', 'kernel': what, 'achieved': gbs, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': gbs / HBM_PEAK_GBS,
                              'avg_launch_ms': ms, 'algorithmic_bytes': byts, 'traffic': None}
        result['kernels'] = [{'kernel': k, 'avg_ms': mean_ms(k)} for k in
                             (kname, 'batch_norms_kernel', 'occ_flags_kernel', 'rowwise_apply_kernel(users)', 'rowwise_apply_kernel(items)', 'sort_ids')
                             if timings.get(k)]
        if int(os.environ.get('CDR_BENCH_SHARED_GPU', '0')):
            result['data'] = 'synthetic; FUNCTIONAL CHECK ONLY: all ranks share cuda:0 over gloo'

    try:                                          # a failure in this extra leg must not cost the headline measurement
        if dim_mode:
            # phase switch: the OVERLAP step and the full-sort run on ROW shards (ids / k candidates are all they exchange there),
            # so the user tables with their Adam moments and the target item table are transposed once (one all-to-all each)
            from types import SimpleNamespace
            from recbole_cdr_amd.dimshard import dim_to_row_shards, state_to_row_shards
            barrier(world)
            t0 = time.perf_counter()
            if dom_groups:
                from recbole_cdr_amd.dimshard import cols_to_row_shards, state_cols_to_row_shards
                holders = {'source': list(range(half)), 'target': list(range(half, world))}
                st = steps.pop(my_dom)
                mine_u, mine_i = st.ustate, tabs[my_dom[0] + 'i']
                tabs.clear()
                del st
                torch.cuda.empty_cache()
                ust = {d: state_cols_to_row_shards(mine_u if d == my_dom else None, n_users, Ds, holders[d], args.opt == 'adam')
                       for d in ('source', 'target')}
                del mine_u
                ti_rows = cols_to_row_shards(mine_i if my_dom == 'target' else None, n_items, Ds, holders['target'])
                del mine_i
            else:
                ust, ti_cols = {}, tabs['ti']
                tabs.clear()
                for d in ('source', 'target'):
                    st = steps.pop(d)
                    ustate = st.ustate
                    del st                                  # gradient / sort buffers and the item moments go first
                    torch.cuda.empty_cache()
                    ust[d] = state_to_row_shards(ustate, consume=True)
                ti_rows = dim_to_row_shards(ti_cols)
                del ti_cols
            barrier(world)
            result['relayout_dim_to_row_s'] = time.perf_counter() - t0
            steps = {d: SimpleNamespace(ustate=ust[d], istate=None) for d in ('source', 'target')}
            tabs = {'su': ust['source'].table, 'tu': ust['target'].table, 'ti': ti_rows}
            del ust, ti_rows                            # `steps` owns the moments now (the full-sort leg below drops them)
            torch.cuda.empty_cache()

        # ---- OVERLAP phase (emcdr.py:133-137): mapping(source_user_e[idx]) -> target_user_e[idx], OB = 65,536 shuffled
        # overlapped ids per rank, linear mapping D x D; the user tables' row-wise Adam state is the one the BPR steps use
        if not getattr(args, 'no_map', False):
            from recbole_cdr_amd.fused import FusedMapStep
            Wm = torch.nn.Parameter(xavier_table(D, D, D, gen, dev))
            if sharded:
                dist.broadcast(Wm.data, 0)                    # the mapping is replicated: same initial weights on every rank
            fmap = FusedMapStep(tabs['su'], tabs['tu'], lambda x: F_.linear(x, Wm, None, B_.ACT_NONE), [Wm], 65536, opt=args.opt,
                                group=(dist.group.WORLD if sharded else None), layers=[(Wm, None, B_.ACT_NONE)],
                                source_state=steps['source'].ustate, target_state=steps['target'].ustate)
            OB = 65536
            # the reference's OverlapDataloader yields slices of a shuffled arange (data/dataloader.py:37-52): distinct ids
            perm_ids = torch.randperm(OU - 1, device=dev, generator=gen)[:4 * OB] + 1
            idxs = [perm_ids[i * OB:(i + 1) * OB].view(-1, 1).contiguous() for i in range(4)]
            uniq = not sharded
            for i in range(10):
                fmap.step(idxs[i % 4], unique=uniq)
            barrier(world)
            if uniq:
                B_.timing_enable(dev, 128)                    # HIP events around the step's first launch, on the launch stream
            t0 = time.perf_counter()
            for i in range(60):
                fmap.step(idxs[i % 4], unique=uniq)
            barrier(world)
            tm = torch.tensor([(time.perf_counter() - t0) / 60], device=dev, dtype=torch.float64)
            kms = None
            if uniq:
                ks = [ms for nm, ms in B_.timing_collect(dev) if nm == 'map_step_kernel']
                B_.timing_enable(dev, 0)
                kms = sum(ks) / len(ks) if ks else None
            if world > 1:
                tm = ctrl_max(tm)
            ob_bytes = OB * 2 * 6 * 4 * D                      # two rows per id, 6 x 4D bytes per row
            result['overlap_phase'] = {'ms_per_step': float(tm) * 1e3, 'overlap_ids_per_s': OB * world / float(tm),
                                       'batch_per_rank': OB, 'mapping': 'linear %dx%d' % (D, D), 'loss': float(fmap.loss),
                                       'launches': 2 if uniq else None,
                                       'roofline': {'bound': 'hbm', 'achieved': ob_bytes / float(tm) / 1e9, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                                                    'frac': ob_bytes / float(tm) / 1e9 / HBM_PEAK_GBS,
                                                    'algorithmic_bytes': ob_bytes, 'what': '6,144 B per id (SURVEY 8d), wall time of the whole step'}}
            if kms:
                # the step's first launch does all of the table traffic (the second reduces the mapping's gradient partials)
                result['overlap_phase']['roofline_kernel'] = {
                    'bound': 'hbm', 'kernel': 'map_pipe_kernel (cdr_map_step_unique, launch 1 of 2)', 'avg_launch_ms': kms,
                    'achieved': ob_bytes / (kms * 1e-3) / 1e9, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                    'frac': ob_bytes / (kms * 1e-3) / 1e9 / HBM_PEAK_GBS, 'algorithmic_bytes': ob_bytes,
                    'traffic': pmc_traffic('map_step_kernel')}
            if uniq:
                small = [perm_ids[i * 100:(i + 1) * 100].view(-1, 1).contiguous() for i in range(4)]
                for i in range(3):
                    fmap.step(small[i % 4], unique=True)
                torch.cuda.synchronize(); t0 = time.perf_counter()
                for i in range(200):
                    fmap.step(small[i % 4], unique=True)
                torch.cuda.synchronize()
                result['overlap_phase']['ob100_ms_per_step'] = (time.perf_counter() - t0) / 200 * 1e3
                # the reference's DEFAULT mapping is the tanh MLP (properties/model/EMCDR.yaml: non_linear, hidden 128): the same two-launch
                # step on its general-shape kernel (the wave-group kernel above is written for the linear mapping)
                W1, b1 = torch.nn.Parameter(xavier_table(128, D, D, gen, dev)), torch.nn.Parameter(torch.zeros(128, device=dev))
                W2, b2 = torch.nn.Parameter(xavier_table(D, 128, 128, gen, dev)), torch.nn.Parameter(torch.zeros(D, device=dev))
                fmlp = FusedMapStep(tabs['su'], tabs['tu'],
                                    lambda x: F_.linear(F_.linear(x, W1, b1, B_.ACT_TANH), W2, b2, B_.ACT_NONE), [W1, b1, W2, b2], 65536,
                                    opt=args.opt, layers=[(W1, b1, B_.ACT_TANH), (W2, b2, B_.ACT_NONE)],
                                    source_state=steps['source'].ustate, target_state=steps['target'].ustate)
                nl = {}
                for name, id_lists, reps in (('OB=65536', idxs, 20), ('OB=100', small, 200)):
                    for i in range(3):
                        fmlp.step(id_lists[i % 4], unique=True)
                    torch.cuda.synchronize(); t0 = time.perf_counter()
                    for i in range(reps):
                        fmlp.step(id_lists[i % 4], unique=True)
                    torch.cuda.synchronize()
                    nl[name + '_ms_per_step'] = (time.perf_counter() - t0) / reps * 1e3
                nl['mapping'] = 'tanh MLP %d-128-%d' % (D, D)
                nl['frac_of_hbm_peak_at_OB=65536'] = ob_bytes / (nl['OB=65536_ms_per_step'] * 1e-3) / 1e9 / HBM_PEAK_GBS
                result['overlap_phase']['non_linear'] = nl
                del fmlp
            del fmap                                   # it shares (and would keep alive) the user tables' Adam moments

    except Exception as e:  # noqa: BLE001
        result.setdefault('leg_errors', {})['relayout_overlap'] = repr(e)[:500]
        print('bench: relayout_overlap leg failed: %r' % (e,), file=sys.stderr)
    # ---- roofline of the dominant kernel(s): algorithmic bytes / HIP-event time of each native call --------------
    if rank == 0 and not sharded:
        med_ms = lambda k: (sorted(timings[k])[len(timings[k]) // 2]) if timings.get(k) else 0.0
        fused = bool(getattr(steps['source'], 'fuse_singles', False))
        nmom = 6 if args.opt == 'adam' else 2
        row_b = 4 * D
        # occupancy statistics of one batch (outside the timed region): distinct rows, rows that occur once, their occurrences
        u, p, n = batches[0]['source']
        cu_ = torch.unique(u, return_counts=True)[1]
        ci_ = torch.unique(torch.cat([p, n]), return_counts=True)[1]
        du, di = int(cu_.numel()), int(ci_.numel())                       # distinct touched rows
        su_, si_ = int((cu_ == 1).sum()), int((ci_ == 1).sum())            # rows with exactly one occurrence (= single occurrences)
        both_single = int(((torch.bincount(p, minlength=n_items)[p] + torch.bincount(n, minlength=n_items)[p] == 1) &
                           (torch.bincount(p, minlength=n_items)[n] + torch.bincount(n, minlength=n_items)[n] == 1)).sum()) if fused else 0
        kernels = []
        if fused:
            # SURVEY 8d bytes (the figure roofline.frac is computed from): 6 x 4D per row the kernel UPDATES (its single-occurrence
            # rows: w, m, v read and written) + 4D per row it only gathers (occurrences of duplicate rows).  design_bytes adds what
            # this implementation moves on top: ids, flags, squared norms, and the compact gradient rows of the duplicate occurrences.
            dup_occ = (B - su_) + (2 * B - si_)
            alg_fa = (su_ + si_) * nmom * row_b + dup_occ * row_b
            des_fa = alg_fa + B * (24 + 4) + (B - su_) * row_b + (B - both_single) * row_b
            alg_du = (B - su_) * row_b + (du - su_) * nmom * row_b          # duplicate users: one gradient row per occurrence + RMW per row
            alg_di = (2 * B - si_) * row_b + (di - si_) * nmom * row_b
            des_du = alg_du + (B - su_) * 8 + (du - su_) * 8
            des_di = alg_di + (2 * B - si_) * 8 + (di - si_) * 8
            des_fl = 3 * B * (4 + 4 + 1)                                    # keys + perm read, flag bytes written
            des_bn = B * (2 * row_b + 16)                                   # the EmbLoss norms: user + positive row of every triple gathered once more
            # (round 4: the duplicate rows of BOTH tables go through one launch, rowwise_apply_dups2_kernel, blockIdx.y = table; its bracket carries
            #  the item side's tag)
            for kname, alg_b, des_b in (('bpr_fwd_apply_kernel', alg_fa, des_fa), ('batch_norms_kernel', 0, des_bn), ('occ_flags_kernel', 0, des_fl),
                                        ('rowwise_apply_kernel(items)', alg_du + alg_di, des_du + des_di)):
                ms = med_ms(kname)
                kernels.append({'kernel': 'rowwise_apply_dups2_kernel [duplicate rows of the user and the item table, one launch]' if kname.startswith('rowwise') else kname,
                                'avg_ms': mean_ms(kname), 'median_ms': ms,
                                'algorithmic_bytes': alg_b, 'design_bytes': des_b,
                                'achieved_GBps': alg_b / (ms * 1e-3) / 1e9 if ms > 0 else 0.0, 'frac': alg_b / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS if ms > 0 else 0.0,
                                'design_GBps': des_b / (ms * 1e-3) / 1e9 if ms > 0 else 0.0})
        else:
            alg = {'bpr_fwd_grad_kernel': (B * 3 * row_b, B * (3 * row_b + 24) + B * 2 * row_b),
                   'rowwise_apply_kernel(users)': (du * nmom * row_b, B * (8 + row_b) + du * nmom * row_b),
                   'rowwise_apply_kernel(items)': (di * nmom * row_b, 2 * B * (8 + row_b) + di * nmom * row_b)}
            for kname, (alg_b, des_b) in alg.items():
                ms = med_ms(kname)
                kernels.append({'kernel': kname, 'avg_ms': mean_ms(kname), 'median_ms': ms, 'algorithmic_bytes': alg_b, 'design_bytes': des_b,
                                'achieved_GBps': alg_b / (ms * 1e-3) / 1e9 if ms > 0 else 0.0, 'frac': alg_b / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS if ms > 0 else 0.0,
                                'design_GBps': des_b / (ms * 1e-3) / 1e9 if ms > 0 else 0.0})
        kernels.append({'kernel': 'sort_ids (make_keys + rocprim radix sort of the 3 B (id, occurrence) pairs of a domain step)',
                        'avg_ms': mean_ms('sort_ids'), 'median_ms': med_ms('sort_ids')})
        dom_k = max(kernels[:-1], key=lambda k: k['median_ms'])
        result['roofline'] = {'bound': 'hbm', 'kernel': dom_k['kernel'], 'achieved': dom_k['achieved_GBps'],
                              'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': dom_k['frac'],
                              'avg_launch_ms': dom_k['median_ms'], 'algorithmic_bytes': dom_k['algorithmic_bytes'],
                              'bytes_model': 'SURVEY 8d: 6 x 4D bytes per row the launch updates (w, m, v read + written) + 4D per row it only gathers; '
                                             'avg_launch_ms = median of the HIP-event brackets of ' + ('the single-stream steps right behind the timed region '
                                             '(in the timed region the two domains\' kernels overlap on two streams: brackets taken there would time a kernel '
                                             'sharing the chip)' if two_streams else 'the timed region\'s launches'),
                              'design_bytes': dom_k['design_bytes'], 'design_GBps': dom_k['design_GBps'],
                              'design_frac': dom_k['design_GBps'] / HBM_PEAK_GBS,
                              'traffic': pmc_traffic(dom_k['kernel'].split(' ')[0]),
                              'rocprof': 'profiles/r05_bench_c5_headline_kernel_stats.csv = rocprofv3 --kernel-trace --stats of `python bench.py --headline-only` '
                                         '(single stream, this kernel on the headline batch only; the default command also launches it on the k = 4 and synthetic-grid '
                                         'batches and beside the other domain\'s kernels, so ITS stats file averages over several situations)'}
        result['kernels'] = kernels
        result['batch_occupancy'] = {'triples': B, 'distinct_user_rows': du, 'distinct_item_rows': di, 'single_user_rows': su_, 'single_item_rows': si_}
        # the STEP against SURVEY 8d's floor for a fused row-wise-Adam step: 6 x 4D bytes per touched row
        dom_ms = dt / args.steps * 1e3 / 2.0
        step_bytes = B * 3 * nmom * row_b
        distinct_bytes = (du + di) * nmom * row_b
        result['roofline_step'] = {'bound': 'hbm', 'what': ('one domain step (batch norms + sort + flags + forward/optimizer + duplicate-row applies) of %d triples against SURVEY 8d\'s '
                                   '9,216 B/triple at D=128 (6 x 4D per touched row, 3 rows per triple, no reuse counted)' % B) +
                                   ('; ms_per_domain_step = half of the two-stream step (the two domain steps overlap)' if two_streams else ''),
                                   'achieved': step_bytes / (dom_ms * 1e-3) / 1e9, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                                   'frac': step_bytes / (dom_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 'ms_per_domain_step': dom_ms,
                                   'algorithmic_bytes': step_bytes,
                                   'distinct_rows': {'algorithmic_bytes': distinct_bytes, 'achieved': distinct_bytes / (dom_ms * 1e-3) / 1e9,
                                                     'frac': distinct_bytes / (dom_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                                     'what': '6 x 4D per DISTINCT touched row (%d user + %d item rows)' % (du, di)},
                                   'traffic': pmc_traffic('domain_step')}
        # north_star's "dual-domain embedding gather": the forward-only gather kernel on the same tables and batches (3 rows + 3 ids per
        # triple, SURVEY 8d: 1,560 B per triple at D = 128), timed on its own after the timed region
        try:
            out4 = torch.zeros(4, device=dev, dtype=torch.float32)
            B_.timing_enable(dev, 64)
            for rep in range(6):
                for dom, (tu, ti) in (('source', ('su', 'si')), ('target', ('tu', 'ti'))):
                    uu, pp, nn = batches[rep % pool][dom]
                    B_.call('cdr_bpr_fwd', B_.ctx(dev), B_.stream(), B_.f32(tabs[tu]), B_.f32(tabs[ti]), D, B_.i64(uu), B_.i64(pp), B_.i64(nn), B,
                            1e-10, 0.01, B_.f32(out4), None)
            torch.cuda.synchronize()
            gat = [ms for nm, ms in B_.timing_collect(dev) if nm == 'bpr_fwd_kernel'][4:]     # in-library HIP events; first launches dropped
            B_.timing_enable(dev, 0)
            gms = sorted(gat)[len(gat) // 2]
            gb = B * (3 * row_b + 24)
            result['roofline_gather'] = {'bound': 'hbm', 'kernel': 'bpr_fwd_kernel (forward-only gather + BPR/EmbLoss sums, one domain batch)', 'achieved': gb / (gms * 1e-3) / 1e9,
                                         'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': gb / (gms * 1e-3) / 1e9 / HBM_PEAK_GBS, 'avg_launch_ms': gms,
                                         'algorithmic_bytes': gb, 'traffic': pmc_traffic('bpr_fwd_kernel'),
                                         'byte_models': {'per_triple_B': 3 * row_b + 24, 'per_positive_reuse_floor_B_at_k1': 3 * row_b,
                                                         'note': 'k = 1: both SURVEY 8d models coincide up to the 24 B of ids; the k = 4 leg (`per_positive_k4`) is '
                                                                 'measured against (2 + k) 4D / k = 768 B per triple'}}
        except Exception as e:  # noqa: BLE001
            result.setdefault('leg_errors', {})['gather'] = repr(e)[:300]

    try:
        # ---- the same step with the two domains on their own HIP streams (they share nothing: disjoint tables and optimizer state):
        # tails and small launches of one domain run under the other's kernels.  Reported beside the headline, which stays the
        # single-stream measurement so that its per-kernel brackets and the rocprofv3 averages are those of kernels running alone.
        if rank == 0 and not sharded and not two_streams and not getattr(args, 'no_extra_legs', False):
            dstreams = {d: torch.cuda.Stream(device=dev) for d in ('source', 'target')}
            for d in dstreams:
                with torch.cuda.stream(dstreams[d]):
                    B_.ctx(dev)                            # the native context of each stream, created outside the timed region

            def two_stream_step(i):
                cur = torch.cuda.current_stream()
                b = batches[i % pool]
                for d in ('source', 'target'):
                    dstreams[d].wait_stream(cur)
                    with torch.cuda.stream(dstreams[d]):
                        steps[d].step(*b[d])
                for d in ('source', 'target'):
                    cur.wait_stream(dstreams[d])
            for i in range(max(args.warmup, 2)):
                two_stream_step(i)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for i in range(args.steps):
                two_stream_step(i)
            torch.cuda.synchronize()
            dt2 = (time.perf_counter() - t0) / args.steps
            result['two_streams'] = {'ms_per_step': dt2 * 1e3, 'value': 2 * B / dt2, 'unit': 'interactions/s', 'steps': args.steps,
                                     'what': 'the SOURCE and the TARGET domain step of every benchmark step enqueued on two HIP streams (same kernels, same batches)'}
    except Exception as e:  # noqa: BLE001
        result.setdefault('leg_errors', {})['two_streams'] = repr(e)[:500]
    try:
        # ---- EMCDR's DEFAULT latent factor model (MF: pointwise MSE, emcdr.py:111-122) on recbole's pointwise layout at the headline's
        # table sizes: S = B / 2 positives + one sampled negative each = B rows per domain step, the per-positive step against the per-row forms
        if rank == 0 and not sharded and not getattr(args, 'no_extra_legs', False) and args.opt == 'adam' and D % 4 == 0 and D <= 256:
            from recbole_cdr_amd.fused import KMajorPointStep, FusedPointStep
            Sp = B // 2
            st0 = steps['source']
            mfb = []
            for _ in range(4):
                u_ = torch.randint(1, OU, (Sp,), device=dev, generator=gen)
                it_ = torch.randint(1 + TOI, 1 + 2 * TOI, (2 * Sp,), device=dev, generator=gen)
                mfb.append((u_.repeat(2), it_, torch.cat([torch.ones(Sp, device=dev), torch.zeros(Sp, device=dev)])))
            mf = {}
            for name, mk in (('per_positive', lambda: KMajorPointStep(tabs['su'], tabs['si'], Sp, k=1, loss='mse', opt='adam', reg_weight=0.01,
                                                                      user_state=st0.ustate, item_state=st0.istate)),
                             ('per_row_two_pass', lambda: FusedPointStep(tabs['su'], tabs['si'], 2 * Sp, loss='mse', opt='adam', reg_weight=0.01,
                                                                         user_state=st0.ustate, item_state=st0.istate, fuse_singles=False))):
                stp = mk()
                for i in range(3):
                    stp.step(*mfb[i % 4])
                torch.cuda.synchronize(); t0 = time.perf_counter()
                n_it = max(args.steps // 2, 5)
                for i in range(n_it):
                    stp.step(*mfb[i % 4])
                torch.cuda.synchronize()
                mf[name] = (time.perf_counter() - t0) / n_it * 1e3
                del stp
            rows_mf = 2 * Sp
            result['mf_pointwise'] = {
                'rows_per_domain_step': rows_mf, 'positives': Sp, 'k': 1, 'ms_per_domain_step': mf['per_positive'], 'rows_per_s': rows_mf / (mf['per_positive'] * 1e-3),
                'per_row_two_pass_ms': mf['per_row_two_pass'], 'speedup_vs_two_pass': mf['per_row_two_pass'] / mf['per_positive'],
                'frac_of_hbm_peak_at_6_x_4D_per_touched_row': (Sp * 6 * 4 * D + 2 * Sp * 6 * 4 * D) / (mf['per_positive'] * 1e-3) / 1e9 / HBM_PEAK_GBS,
                'what': 'EMCDR-MF (pointwise MSE + EmbLoss) row-wise Adam step on recbole\'s pointwise batch layout (users tiled 1 + k times, '
                        'items = [positives | negatives]) at the headline\'s table sizes: fused.KMajorPointStep (cdr_point_step_fused_kmajor) against '
                        'the round-2 two-pass form; bytes model: 6 x 4D per distinct row touched (S user rows + 2 S item rows)'}
            del mfb
            torch.cuda.empty_cache()
    except Exception as e:  # noqa: BLE001
        result.setdefault('leg_errors', {})['mf_pointwise'] = repr(e)[:500]
    try:
        # ---- the per-positive (k-major) step at k = 4, and the reference-default 2,048-row batch as one hipGraph -------------
        if rank == 0 and not sharded and not getattr(args, 'no_extra_legs', False):
            from recbole_cdr_amd.fused import KMajorBPRStep
            k = 4
            S = B // k
            st0 = steps['source']
            km = KMajorBPRStep(tabs['su'], tabs['si'], S, k=k, opt=args.opt, reg_weight=0.01, user_state=st0.ustate, item_state=st0.istate)
            kb = [(torch.randint(1, OU, (S,), device=dev, generator=gen), torch.randint(1 + TOI, 1 + 2 * TOI, (S,), device=dev, generator=gen),
                   torch.randint(1 + TOI, 1 + 2 * TOI, (B,), device=dev, generator=gen)) for _ in range(4)]
            tiled = [(b[0].repeat(k), b[1].repeat(k), b[2]) for b in kb]
            def timed(fn, data, n=20):
                for i in range(3):
                    fn(*data[i % 4])
                torch.cuda.synchronize(); t0 = time.perf_counter()
                for i in range(n):
                    fn(*data[i % 4])
                torch.cuda.synchronize()
                return (time.perf_counter() - t0) / n * 1e3
            B_.timing_enable(dev, 256)
            ms_k = timed(km.step, kb)
            kt = {}
            for nm, ms in B_.timing_collect(dev):
                kt.setdefault(nm, []).append(ms)
            B_.timing_enable(dev, 0)
            ms_t = timed(st0.step, tiled)
            # median of the launches after the three warm-up calls: a bracket whose first event is recorded on an idle stream also spans the
            # host's launch latency (round 2's figure averaged those in and read 1.8x the rocprofv3 duration of the same launches)
            fk = sorted(kt.get('bpr_fwd_kmajor_kernel', [0.0])[3:]) or [0.0]
            fwd_ms = fk[len(fk) // 2]
            fwd_bytes = S * ((2 + k) * 4 * D + 8 * (2 + k)) + S * 4 * D + (S + B) * 8       # rows + ids read, GU rows + item records written
            result['per_positive_k4'] = {
                'rows_per_domain_step': B, 'k': k, 'ms_per_domain_step': ms_k, 'rows_per_s': B / (ms_k * 1e-3),
                'per_triple_step_same_batch_ms': ms_t, 'speedup_vs_per_triple_step': ms_t / ms_k}
            if getattr(km, 'fuse_singles', False):
                # the per-positive forward that also updates the rows occurring once (cdr_bpr_step_fused_kmajor): SURVEY 8d bytes = 6 x 4D per
                # row it updates + 4D per row it only gathers, from the single-occurrence flags of the last step
                fstride = (2 + k + 3) // 4 * 4
                fl = km.flags[:S * fstride].view(S, fstride)[:, :2 + k].float().sum(0)
                n_single = float(fl.sum()); n_rows = float(S * (2 + k))
                fk2 = sorted(kt.get('bpr_fwd_apply_kernel', [0.0])[3:]) or [0.0]
                fa_ms = fk2[len(fk2) // 2]
                fa_bytes = n_single * 6 * 4 * D + (n_rows - n_single) * 4 * D
                result['per_positive_k4']['forward_optimizer_kernel'] = {
                    'kernel': 'bpr_fwd_apply_kmajor_kernel', 'median_ms': fa_ms, 'algorithmic_bytes': fa_bytes, 'single_rows': n_single,
                    'rows_gathered': n_rows, 'achieved_GBps': fa_bytes / (fa_ms * 1e-3) / 1e9 if fa_ms > 0 else 0.0,
                    'frac': fa_bytes / (fa_ms * 1e-3) / 1e9 / HBM_PEAK_GBS if fa_ms > 0 else 0.0,
                    'byte_model': 'SURVEY 8d: 6 x 4D per row the launch updates + 4D per row it only gathers; (2 + k) rows per positive'}
            else:
                result['per_positive_k4']['gather_kernel'] = {
                    'kernel': 'bpr_fwd_kmajor_kernel', 'avg_ms': fwd_ms, 'algorithmic_bytes': fwd_bytes,
                    'achieved_GBps': fwd_bytes / (fwd_ms * 1e-3) / 1e9 if fwd_ms > 0 else 0.0,
                    'frac': fwd_bytes / (fwd_ms * 1e-3) / 1e9 / HBM_PEAK_GBS if fwd_ms > 0 else 0.0,
                    'byte_model': '(2 + k) rows + (2 + k) ids read, 1 gradient row + (1 + k) 8-B records written per positive = '
                                  '%d B per triple at k = 4' % (fwd_bytes // B)}
            del km
            # ---- SURVEY 8d's synthetic grid on the per-triple step: B = 65,536 uniform; B = 1,048,576 and 65,536 with Zipf(1.05) positives
            def zipf_items(nn, lo):
                r = torch.rand(nn, device=dev, generator=gen, dtype=torch.float64)
                a = 1.05
                x = ((float(TOI) ** (1 - a) - 1) * r + 1) ** (1 / (1 - a))            # inverse CDF of the continuous Zipf(1.05) over TOI ranks
                return lo + (x.long().clamp_(1, TOI) - 1)
            grid = {}
            for name, nb, zipf in (('B=65536 uniform', 65536, False), ('B=1048576 zipf(1.05) positives', B, True), ('B=65536 zipf(1.05) positives', 65536, True)):
                gs = FusedBPRStep(tabs['su'], tabs['si'], nb, opt=args.opt, reg_weight=0.01, user_state=st0.ustate, item_state=st0.istate)
                gb_ = [(torch.randint(1, OU, (nb,), device=dev, generator=gen),
                        zipf_items(nb, 1 + TOI) if zipf else torch.randint(1 + TOI, 1 + 2 * TOI, (nb,), device=dev, generator=gen),
                        torch.randint(1 + TOI, 1 + 2 * TOI, (nb,), device=dev, generator=gen)) for _ in range(4)]
                ms_g = timed(gs.step, gb_, n=40 if nb < B else 20)
                top = int(torch.bincount(gb_[0][1] - (1 + TOI)).max()) if zipf else None
                grid[name] = {'rows_per_domain_step': nb, 'ms_per_domain_step': ms_g, 'rows_per_s': nb / (ms_g * 1e-3),
                              'frac_of_hbm_peak_at_9216_B_per_triple': nb * 3 * 6 * 4 * D / (ms_g * 1e-3) / 1e9 / HBM_PEAK_GBS}
                if getattr(gs, 'device_counts', False) and nb <= 65536:         # (larger sorts are not replay-safe: EMCDR.fused_graph_key)
                    # the same step replayed as a hipGraph (update counts on the device: cdr_bpr_step_fused_dev) -- what CrossDomainTrainer's
                    # rowwise loop does on a device loader: the ~20 launches of a step without their launch gaps
                    side_ = torch.cuda.Stream(device=dev)
                    with torch.cuda.stream(side_):
                        gs.step(*gb_[0])
                    torch.cuda.synchronize()
                    cg = torch.cuda.CUDAGraph()
                    with B_.capturing(cg, side_):
                        gs.step(*gb_[0])

                    def rep(*_a):
                        cg.replay(); gs.replayed()
                    ms_r = timed(rep, gb_, n=40 if nb < B else 20)
                    grid[name]['hipgraph_ms_per_domain_step'] = ms_r
                    grid[name]['hipgraph_frac_of_hbm_peak_at_9216_B_per_triple'] = nb * 3 * 6 * 4 * D / (ms_r * 1e-3) / 1e9 / HBM_PEAK_GBS
                    del cg
                if top is not None:
                    grid[name]['occurrences_of_hottest_item'] = top
                del gs, gb_
            result['synthetic_grid'] = grid
            # the reference's default train_batch_size (2,048 rows, properties/overall.yaml:19): 4 launches, replayed as a hipGraph
            S2 = 2048
            sm = KMajorBPRStep(tabs['su'], tabs['si'], S2, k=1, opt=args.opt, reg_weight=0.01, user_state=st0.ustate, item_state=st0.istate)
            sb = [(torch.randint(1, OU, (S2,), device=dev, generator=gen), torch.randint(1 + TOI, 1 + 2 * TOI, (S2,), device=dev, generator=gen),
                   torch.randint(1 + TOI, 1 + 2 * TOI, (S2,), device=dev, generator=gen)) for _ in range(4)]
            ms_e = timed(sm.step, sb, n=200)
            sm.capture(S2)
            ms_g = timed(sm.replay, sb, n=200)
            ms_i = timed(lambda *a: sm.replay(), sb, n=200)
            result['small_batch_2048'] = {'rows_per_domain_step': S2, 'eager_ms': ms_e, 'hipgraph_ms_with_3_id_copies': ms_g,
                                          'hipgraph_ms_ids_written_in_place': ms_i, 'rows_per_s': S2 / (ms_i * 1e-3),
                                          'launches': 4, 'note': 'C5 table sizes; {forward || rank-count}, {rank-scatter || loss finish}, '
                                                                 'item apply, user apply; Adam update counts on the device'}
            del sm
    except Exception as e:  # noqa: BLE001
        result.setdefault('leg_errors', {})['per_positive_small_batch'] = repr(e)[:500]
        print('bench: per-positive / small-batch leg failed: %r' % (e,), file=sys.stderr)

    try:                                          # a failure in this extra leg must not cost the headline measurement
        # ---- metric 2: full-sort items scored / s (emcdr.py:208-233, TARGET phase) at the reference's U and at U=1024 --
        if rank == 0 and not sharded and not args.no_fullsort:
            fs = {}
            slab = tabs['ti'][:1 + TOI]
            # the optimizer state is not needed any more; make room for the [U, N] score matrix (40 GB at U=1024)
            for st in steps.values():
                st.ustate = st.istate = None
                st.GU = st.GP = None
            torch.cuda.empty_cache()
            for Uu in (1, 1024):
                ue = tabs['tu'][1:1 + Uu].contiguous()
                reps = 5 if Uu == 1 else 3
                out = torch.empty(Uu, slab.shape[0], device=dev, dtype=torch.float32)
                F_.fullsort_scores(ue, slab, out=out)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(reps):
                    F_.fullsort_scores(ue, slab, out=out)
                e1.record(); torch.cuda.synchronize()
                ms = e0.elapsed_time(e1) / reps
                N = slab.shape[0]
                byts = 4.0 * N * D + 4.0 * Uu * D + 4.0 * Uu * N
                flops = 2.0 * Uu * N * D
                fs['U=%d' % Uu] = {'items_per_s': Uu * N / (ms * 1e-3), 'ms': ms, 'N': N,
                                   'achieved_GBps': byts / (ms * 1e-3) / 1e9, 'hbm_frac': byts / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                   'achieved_TFLOPs': flops / (ms * 1e-3) / 1e12,
                                   'mfma_frac': flops / (ms * 1e-3) / 1e12 / FP32_MFMA_PEAK_TFLOPS}
                del out
                # the evaluation recbole actually runs on those scores: mask (PAD + 50 history columns per user) + top-10, fused
                # after the contraction so that the [U, N] matrix is never written
                hist = torch.sort(torch.randint(1, N, (Uu, 50), device=dev, generator=gen), dim=1).values.reshape(-1).contiguous()
                hptr = torch.arange(0, Uu + 1, device=dev, dtype=torch.int64) * 50
                F_.fullsort_topk(ue, slab, None, k=10, hist_indptr=hptr, hist_cols=hist)
                torch.cuda.synchronize()
                e0.record()
                for _ in range(reps):
                    F_.fullsort_topk(ue, slab, None, k=10, hist_indptr=hptr, hist_cols=hist)
                e1.record(); torch.cuda.synchronize()
                mk = e0.elapsed_time(e1) / reps
                fs['U=%d' % Uu]['masked_top10'] = {'ms': mk, 'items_per_s': Uu * N / (mk * 1e-3),
                                                   'achieved_TFLOPs': flops / (mk * 1e-3) / 1e12}
            result['fullsort'] = fs
        # ---- metric 2 at N > 1: the target item table is row-sharded; every rank scores its rows, the [U, N/G] partials are
        # all-gathered and re-ordered into the reference's [U, N] layout on every rank (shard.ShardedFullSort) ------------
        if sharded and not args.no_fullsort:
            from recbole_cdr_amd.shard import ShardedFullSort
            for st in steps.values():
                st.ustate = st.istate = None
            torch.cuda.empty_cache()
            fsr = ShardedFullSort(tabs['ti'], 1 + TOI)
            fs = {}
            for Uu in (1, 1024):
                ids = torch.arange(1, 1 + Uu, device=dev, dtype=torch.int64)
                ue = fsr.user_rows(tabs['tu'], ids)
                out = fsr.scores(ue); del out
                reps = 3
                barrier(world)
                t0 = time.perf_counter()
                for _ in range(reps):
                    out = fsr.scores(ue); del out
                barrier(world)
                t_all = torch.tensor([(time.perf_counter() - t0) / reps], device=dev, dtype=torch.float64)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(reps):
                    part = fsr.local_scores(ue); del part
                e1.record(); torch.cuda.synchronize()
                t_loc = torch.tensor([e0.elapsed_time(e1) / reps * 1e-3], device=dev, dtype=torch.float64)
                if world > 1:
                    t_all, t_loc = ctrl_max(t_all), ctrl_max(t_loc)
                N = 1 + TOI
                hist = torch.sort(torch.randint(1, 1 + TOI, (Uu, 50), device='cpu', generator=torch.Generator().manual_seed(7)),
                                  dim=1).values.reshape(-1).contiguous().to(dev)                 # replicated: same on every rank
                hptr = torch.arange(0, Uu + 1, device=dev, dtype=torch.int64) * 50
                fsr.topk(ue, 10, hist_indptr=hptr, hist_cols=hist)
                barrier(world)
                t0 = time.perf_counter()
                for _ in range(reps):
                    fsr.topk(ue, 10, hist_indptr=hptr, hist_cols=hist)
                barrier(world)
                t_top = torch.tensor([(time.perf_counter() - t0) / reps], device=dev, dtype=torch.float64)
                if world > 1:
                    t_top = ctrl_max(t_top)
                fs['U=%d' % Uu] = {'items_per_s': Uu * N / float(t_all), 'ms': float(t_all) * 1e3, 'N': N,
                                   'masked_top10': {'ms': float(t_top) * 1e3, 'items_per_s': Uu * N / float(t_top),
                                                    'exchange_bytes_per_rank': 12.0 * Uu * 10 * (world - 1)},
                                   'local_scoring_ms': float(t_loc) * 1e3,
                                   'local_scoring_items_per_s_all_ranks': Uu * N / float(t_loc),
                                   'allgather_bytes_per_rank': 4.0 * Uu * fsr.Nl * (world - 1)}
            result['fullsort'] = fs
    except Exception as e:  # noqa: BLE001
        result.setdefault('leg_errors', {})['fullsort'] = repr(e)[:500]
        print('bench: fullsort leg failed: %r' % (e,), file=sys.stderr)
    if world == 1 and not sharded:
        # ... and once more after every other leg has trained on the same tables (OVERLAP steps, k-major, Zipf and small batches, graphs):
        # all of them run a fixed number of steps, so this too is the same in every invocation with the same flags
        result['state_checksum_after_all_legs'] = {k_: repr(exact_sum(v_)) for k_, v_ in tabs.items()}
    return result


_gather_bw_cache = {}


def measured_gather_bandwidth(dev, footprint_bytes, D, B=1 << 20, reps=30):
    """GB/s of a pure random-row gather (cdr_embloss_fwd: squared norms of B + B rows of two tables, 3 floats written) over a working set of
    ``footprint_bytes``: the roof of the cache-resident configurations (C1 / C2 / C4 tables live in L2 / Infinity Cache, where the 8 TB/s
    HBM peak bounds nothing).  Same measurement as tools/mb_cache_gather.py, run here on the config's own footprint."""
    key = (int(footprint_bytes) >> 20, D)
    if key in _gather_bw_cache:
        return _gather_bw_cache[key]
    from recbole_cdr_amd import binding as B_
    rows = max(int(footprint_bytes) // (2 * 4 * D), 16)
    g = torch.Generator(device=dev); g.manual_seed(1)
    U = torch.randn(rows, D, device=dev, generator=g); I = torch.randn(rows, D, device=dev, generator=g)
    u = torch.randint(0, rows, (B,), device=dev, generator=g); i = torch.randint(0, rows, (B,), device=dev, generator=g)
    out3 = torch.empty(3, device=dev)
    call = lambda: B_.call('cdr_embloss_fwd', B_.ctx(dev), B_.stream(), B_.f32(U), B_.f32(I), D, B_.i64(u), B_.i64(i), B, B_.f32(out3))
    for _ in range(5):
        call()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        call()
    e1.record(); torch.cuda.synchronize()
    gbs = 2.0 * B * D * 4 / (e0.elapsed_time(e1) / reps * 1e-3) / 1e9
    # ... and the rate at which the same working set STREAMS through the cache level it sits in (a sum over it, repeated): an upper
    # bound for any access pattern at that footprint -- random 256-B rows are per-request bound well below it, sorted adjacency lists
    # (C4's SpMM) sit in between
    flat = torch.cat([U.reshape(-1), I.reshape(-1)])
    for _ in range(3):
        flat.sum()
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        flat.sum()
    e1.record(); torch.cuda.synchronize()
    stream = flat.numel() * 4.0 / (e0.elapsed_time(e1) / reps * 1e-3) / 1e9
    _gather_bw_cache[key] = (gbs, stream)
    return gbs, stream


def cache_roof(footprint_bytes, row_bytes):
    """The measured read rate out of a working set of this size (tools/mb_cache_bw.hip, a standalone HIP program: uniformly random
    256-B / 512-B row gathers with 8 rows in flight per lane group, and a float4 stream, by footprint; filed as
    profiles/r04_mb_cache_bw.txt by the builder's run of tools/profile_r04.sh -- NOT measured in this invocation).  Returns
    (GB/s of the random-row gather at the first filed footprint >= this one, the filed record) or (None, None)."""
    try:
        recs = [json.loads(l) for l in open(os.path.join(ROOT, 'profiles', 'r04_mb_cache_bw.txt')) if l.strip().startswith('{')]
    except Exception:
        return None, None
    key = 'random_512B_row_gather_GBps' if row_bytes >= 512 else 'random_256B_row_gather_GBps'
    for r in recs:
        if r['footprint_MB'] * 1e6 >= footprint_bytes:
            return float(r[key]), r
    return (float(recs[-1][key]), recs[-1]) if recs else (None, None)


def pmc_traffic(kernel):
    """HBM bytes per launch from the committed rocprofv3 --pmc passes (profiles/pmc_traffic.json), with their provenance: the
    counters are collected in separate profiler runs of this same command (tools/profile_bench.sh), NOT in this invocation."""
    path = os.path.join(ROOT, 'profiles', 'pmc_traffic.json')
    try:
        with open(path, 'rb') as f:
            raw = f.read()
        doc = json.loads(raw)
        v = doc.get(kernel)
    except Exception:
        v = None
    if v is None:
        return None
    import hashlib
    blob = hashlib.sha1(b'blob %d\0' % len(raw) + raw).hexdigest()[:12]          # = `git hash-object profiles/pmc_traffic.json`: a stale copy is visible
    return {'bytes': v, 'collected': doc.get('_collected'), 'git_blob': blob,
            'source': 'profiles/pmc_traffic.json blob %s collected %s (builder run of tools/profile_r%02d.sh: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE '
                      'passes over `python bench.py --no-cpu-baseline --no-fullsort --no-config-legs --no-e2e --no-ingest --single-stream --steps 3 --warmup 1`; '
                      'not measured in this invocation)' % (blob, doc.get('_collected', '?'), int(doc.get('_round', 5)))}


# ------------------------------------------------------------------------------------------------------ C3 / C4 workloads
def run_model_workload(args, world, rank, dev):
    """BASELINE configs[2] (CoNet, Amazon-Books -> Movies sizes, D=128, k=4 pointwise) and configs[3] (BiTGCF, Douban sizes,
    2 layers, D=64) through the drop-in class contract: calculate_loss -> backward -> dense Adam (the reference's
    loop, trainer.py:59-73).  Reports rows/s (a row = one (u, i, label) row of a domain's batch) and the oracle on the
    host cores beside it."""
    import numpy as np
    from recbole_cdr_amd.data.synthetic import SyntheticCrossDomainDataset
    from recbole_cdr_amd.trainer.trainer import DenseAdam
    cfg = {'source_domain': {'NEG_PREFIX': 'neg_'}, 'target_domain': {'NEG_PREFIX': 'neg_'}, 'device': dev}
    pairwise = False
    if args.workload == 'c1':
        from recbole_cdr_amd.model.cross_domain_recommender.cmf import CMF as Model
        # BASELINE configs[0]: the reference's default run (CMF, ml-1m -> ml-100k, embedding_size 64); same id space as C2
        ds = SyntheticCrossDomainDataset(OU=1, TOU=943, SOU=6040, OI=1604, TOI=61, SOI=2280,
                                         n_source_inter=575000, n_target_inter=82000)
        cfg.update(embedding_size=64, alpha=0.5, **{'lambda': 0.0, 'gamma': 0.0})
        S, k = 1024, 1
        name = 'C1: CMF ml-1m->ml-100k sizes (6,984 users x 3,945 items union, shared tables), D=64, 2 x 2,048 rows per step'
    elif args.workload == 'c2':
        from recbole_cdr_amd.model.cross_domain_recommender.emcdr import EMCDR as Model
        # ml-1m -> ml-100k is an ITEM-overlap pair: 1,603 shared titles, no shared users
        ds = SyntheticCrossDomainDataset(OU=1, TOU=943, SOU=6040, OI=1604, TOI=61, SOI=2280,
                                         n_source_inter=575000, n_target_inter=82000)
        cfg.update(latent_factor_model='BPR', source_embedding_size=64, target_embedding_size=64, reg_weight=0.01,
                   mapping_function='non_linear', mlp_hidden_size=[128])
        S, k, pairwise = 2048, 1, True
        name = 'C2: EMCDR-BPR ml-1m->ml-100k sizes (6,984 users x 3,945 items union), D=64, B=2,048, SOURCE-phase steps'
    elif args.workload == 'c3':
        from recbole_cdr_amd.model.cross_domain_recommender.conet import CoNet as Model
        ds = SyntheticCrossDomainDataset(OU=5983, TOU=20986, SOU=129127, OI=1, TOI=18563, SOI=115172,
                                         n_source_inter=400000, n_target_inter=200000)
        cfg.update(embedding_size=128, reg_weight=0.01, mlp_hidden_size=[64, 32, 16, 8])
        S, k, name = 819, 4, 'C3: CoNet Amazon-Books->Movies sizes (156,096 users x 133,736 items), D=128, [256,64,32,16,8], k=4'
    else:
        from recbole_cdr_amd.model.cross_domain_recommender.bitgcf import BiTGCF as Model
        ds = SyntheticCrossDomainDataset(OU=15435, TOU=6607, SOU=2651, OI=1, TOI=25802, SOI=33067,
                                         n_source_inter=809248, n_target_inter=2040000)
        cfg.update(embedding_size=64, n_layers=2, reg_weight=0.001, lambda_source=0.8, lambda_target=0.8, drop_rate=0.3,
                   connect_way='concat', bitgcf_sparse_last_layer=not args.full_last_layer)
        S, k, name = 2048, 1, 'C4: BiTGCF Douban-Book->Movie sizes (24,693 users x 58,870 items, nnz 2x%d / 2x%d), D=64, 2 layers, ' \
                             'full-graph propagation every step' % (len(ds.s_pairs), len(ds.t_pairs))
    torch.manual_seed(2022)
    model = Model(cfg, ds).to(dev)
    # Optimizer: the reference's Adam over every parameter -- built by the product's trainer at N = 1 (for CoNet its deferred row-wise
    # form, lazyadam.DeferredRowAdam: bit-identical to the dense sweep; --dense-adam keeps the literal O(table) sweep)
    deferred, opt = False, None
    rng = np.random.RandomState(2022)
    if pairwise:
        model.set_phase('SOURCE')
        batches = [ds.pairwise_batch('source', S, k, rng, dev) for _ in range(4)]
        rows_per_step = S * k
    else:
        batches = [dict(ds.pointwise_batch('source', S, k, rng, dev), **ds.pointwise_batch('target', S, k, rng, dev)) for _ in range(4)]
        rows_per_step = 2 * S * (1 + k)

    sdp = None
    rowshard = None
    trainer = None
    via = None
    attempts = None
    replicas = False
    if world > 1 or (args.workload == 'c4' and args.force_shard):
        # N > 1 (recbole_cdr_amd/preflight.py: each candidate is built and driven through its first step under a watchdog, verdicts agreed over
        # the gloo control group; none comes up -> every rank runs the one-GPU trainer path on its own GPU, independent replicas):
        #   'rowshard' (c4, BASELINE configs[3] as named): the tables (with their Adam state), the adjacency rows and the transfer-layer degrees
        #       row-sharded over the ranks, per-layer all-gather of E forward and of g (1 + E) backward (bitgcf_shard.py).  The batch is
        #       replicated, so the total work is fixed as N grows: STRONG scaling of the per-step propagation.
        #   'replica-dp': data parallel, every rank its own batches, ONE reduce-scatter + ONE all-gather of the flat parameter buffer per
        #       step and the dense Adam sweep split over the ranks (dp.ShardedDataParallel)
        from recbole_cdr_amd import preflight
        rng = np.random.RandomState(2022 + rank)
        if pairwise:
            dp_batches = [ds.pairwise_batch('source', S, k, rng, dev) for _ in range(4)]
        else:
            dp_batches = [dict(ds.pointwise_batch('source', S, k, rng, dev), **ds.pointwise_batch('target', S, k, rng, dev)) for _ in range(4)]

        # a communicator of its own per candidate, created up front by every rank (a candidate abandoned half-way leaves collectives pending
        # in ITS group only: the next candidate is not paired with them)
        import torch.distributed as dist
        cand_groups = {nm: (dist.new_group(list(range(world))) if world > 1 else None) for nm in ('rowshard', 'replica-dp')}

        def build(name):
            if name == 'rowshard':
                from recbole_cdr_amd.bitgcf_shard import ShardedBiTGCF, NativeGraphOps
                rs = ShardedBiTGCF(ds.num_total_user, ds.num_total_item, ds.num_overlap_user, ds.num_overlap_item, ds.s_pairs, ds.t_pairs,
                                   cfg['embedding_size'], cfg['n_layers'], cfg['lambda_source'], cfg['lambda_target'], cfg['connect_way'],
                                   cfg['reg_weight'], NativeGraphOps(dev), group=cand_groups['rowshard'], drop_rate=cfg['drop_rate'],
                                   batch_loss='replicated' if args.replicated_batch_loss else ('routed' if args.routed_batch_loss else 'auto'))
                return ('rowshard', rs, DenseAdam(list(rs.params.values()), lr=1e-3))
            from recbole_cdr_amd.dp import ShardedDataParallel
            return ('replica-dp', ShardedDataParallel(model, group=cand_groups['replica-dp'], lr=1e-3), None)

        def first_step(c):
            if c[0] == 'rowshard':
                c[2].zero_grad(set_to_none=True); c[1].loss_and_grads(batches[0]); c[2].step()
            else:
                c[1].step(dp_batches[0])
            torch.cuda.synchronize()
        cands = (['rowshard'] if args.workload == 'c4' and not args.replica_dp else []) + ['replica-dp']
        if world > 1:
            lname, c, attempts = preflight.try_layouts(cands[:1] if args.no_layout_fallback else cands, build, first_step, CTRL,
                                                      seconds=args.preflight_seconds, device=dev)
        else:
            lname, c = cands[0], build(cands[0])
        if lname == 'rowshard':
            rowshard, opt = c[1], c[2]
            del model
            torch.cuda.empty_cache()
            model = None
        elif lname == 'replica-dp':
            sdp, batches = c[1], dp_batches
        else:
            if args.no_layout_fallback:
                raise LayoutUnavailable(attempts)
            replicas = True
    if rowshard is None and sdp is None:
        # N = 1: THE PRODUCT'S LOOP.  CrossDomainTrainer.fit over device-resident loaders of this synthetic dataset (the interaction
        # lists tiled to args.steps full batches per epoch, shuffled every epoch, negatives drawn by the device sampler): the trainer
        # captures producer + calculate_loss + backward + Adam once per phase and an epoch is a run of hipGraph replays
        # (trainer/trainer.py::_train_epoch_graphed).  One fit() = one epoch = exactly args.steps steps; the first fit() is the
        # warm-up (it also pays the capture), the second is timed.  Nothing of the step lives in this file any more.
        from recbole_cdr_amd.data import CrossDomainDataloader, OverlapDataloader, DomainTrainLoader
        from recbole_cdr_amd.sampler import DeviceNegSampler
        from recbole_cdr_amd.trainer import CrossDomainTrainer
        from recbole_cdr_amd.utils import InputType
        it_ = InputType.PAIRWISE if pairwise else InputType.POINTWISE
        times = k if pairwise else 1 + k

        def stream_of(pairs, n):
            reps = (n + len(pairs) - 1) // len(pairs)
            sel = np.tile(pairs, (reps, 1))[:n]
            return torch.from_numpy(np.ascontiguousarray(sel[:, 0])).to(dev), torch.from_numpy(np.ascontiguousarray(sel[:, 1])).to(dev)
        su, si = stream_of(ds.s_pairs, args.steps * S)
        tu, ti = stream_of(ds.t_pairs, args.steps * S)
        gen = torch.Generator(device=dev); gen.manual_seed(2022)
        loaders = CrossDomainDataloader(
            DomainTrainLoader({'source_user_id': su, 'source_item_id': si}, 'source_user_id', 'source_item_id', 'source_label', 'neg_',
                              S * times, k, it_, DeviceNegSampler(ds, 'source', ds.s_pairs, dev), shuffle=True, generator=gen),
            DomainTrainLoader({'target_user_id': tu, 'target_item_id': ti}, 'target_user_id', 'target_item_id', 'target_label', 'neg_',
                              S * times, k, it_, DeviceNegSampler(ds, 'target', ds.t_pairs, dev), shuffle=True, generator=gen),
            OverlapDataloader(max(ds.num_overlap_user, ds.num_overlap_item), 100, device=dev, shuffle=True, generator=gen))
        phase = 'SOURCE' if pairwise else 'BOTH'
        tcfg = dict(cfg, learning_rate=1e-3, train_modes=[phase], epoch_num=['1'], epochs=1, eval_step=0, source_split=False,
                    graph_step=not args.no_graph, deferred_adam=not args.dense_adam)
        if os.environ.get('CDR_GRAPH_PIPELINE'):               # A/B runs (tools/, tests): '0' plain order, 'two_ahead', '1' the default
            gp = os.environ['CDR_GRAPH_PIPELINE']
            tcfg['graph_pipeline'] = {'0': False, '1': True}.get(gp, gp)
        if os.environ.get('CDR_GRAPH_UNROLL'):
            tcfg['graph_unroll'] = int(os.environ['CDR_GRAPH_UNROLL'])
        trainer = CrossDomainTrainer(tcfg, model)
        opt = trainer.optimizer
        deferred = type(opt).__name__ == 'RowAwareAdam'
        via = 'CrossDomainTrainer.fit'

    def one_step(i):
        if rowshard is not None:
            opt.zero_grad(set_to_none=True)
            ls, lt = rowshard.loss_and_grads(batches[i % 4])
            opt.step()
            return ls + lt
        return sdp.step(batches[i % 4])

    if trainer is not None:
        trainer.fit(loaders)                                   # warm-up epoch: args.steps steps (>= --warmup), captures the step
        warm = dict(trainer.graph_stats)
        barrier(world)
        t0 = time.perf_counter()
        trainer.fit(loaders)                                   # the timed epoch: exactly args.steps steps + the epoch's shuffle and loss read-back
        barrier(world)
        dt = time.perf_counter() - t0
        loss = torch.tensor(trainer.train_loss_dict[0] / args.steps)
        stats = {k_: trainer.graph_stats[k_] - warm[k_] for k_ in warm}
        assert args.no_graph or stats['replayed'] + stats['eager'] == args.steps, stats
    else:
        for i in range(args.warmup):
            one_step(i)
        barrier(world)
        t0 = time.perf_counter()
        for i in range(args.steps):
            loss = one_step(i)
        barrier(world)
        dt = time.perf_counter() - t0
    if world > 1:
        import torch.distributed as dist
        dt = float(ctrl_max(torch.tensor([dt], device=dev, dtype=torch.float64)))
    job_rows = rows_per_step * (1 if rowshard is not None else world)
    if replicas:
        name = name + '; N = %d INDEPENDENT REPLICAS (fallback: no sharded / data-parallel layout came up on every rank)' % world
    result = {'metric': 'training interactions/sec', 'value': job_rows * args.steps / dt, 'unit': 'interactions/s',
              'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': dt / args.steps * 1e3,
              'higher_is_better': True, 'scaling': 'strong' if rowshard is not None else 'weak', 'vs_baseline': None, 'dtype': 'f32',
              'data': 'synthetic' + ('; FUNCTIONAL CHECK ONLY: all ranks share cuda:0 over gloo' if int(os.environ.get('CDR_BENCH_SHARED_GPU', '0')) else ''),
              'config': {'workload': name + (', drop-in autograd + dense Adam (torch.optim.Adam semantics, update term on 1-ulp rcp / sqrt: DESIGN 5) evaluated lazily per row (bit-identical to this library\'s dense sweep)' if deferred else ', drop-in autograd + dense Adam (torch.optim.Adam semantics, update term on 1-ulp rcp / sqrt: DESIGN 5)') + (', tables + Adam state + adjacency rows row-sharded over %d ranks, per-layer all-gather of E (forward) and of g(1+E) (backward), batch replicated' % world if rowshard is not None else ', data parallel with row-sharded optimizer state (reduce-scatter + all-gather per step)' if sdp is not None else ', eager trainer loop' if args.no_graph else ', batch production + step replayed as one hipGraph per batch'),
                         'rows_per_step': rows_per_step, 'via': via,
                         'trainer_steps': None if trainer is None else dict(stats, warmup_steps_run=args.steps, optimizer=type(opt).__name__,
                                                                            timed='one fit() = one shuffled epoch of exactly --steps full batches, sampler + loader + step + loss read-back')},
              'final_loss': float(loss.sum())}
    if world == 1 and trainer is not None:
        # the trained state as exact fp64 sums (after the timed region): two invocations with the same flags must print the same digits
        if hasattr(model, 'sync_tables'):
            model.sync_tables()
        with torch.no_grad():
            result['state_checksum'] = {n_: repr(exact_sum(p_)) for n_, p_ in model.named_parameters()}
            result['state_checksum']['abs_total'] = repr(float(sum(p_.detach().double().abs().sum() for p_ in model.parameters())))
    if attempts is not None:
        result['layout_fallback'] = {'used': 'rowshard' if rowshard is not None else 'replica-dp' if sdp is not None else 'replicas', 'attempts': attempts,
                                     'fell_back': len(attempts) > 1 or replicas}
    # ---- roofline of the step (SURVEY 8d figures; C1-C4 tables sit in L2 / Infinity Cache, so the HBM fractions are nominal) ----
    step_s = dt / args.steps
    D = cfg['embedding_size'] if 'embedding_size' in cfg else cfg['source_embedding_size']
    nu, ni = ds.num_total_user, ds.num_total_item
    if args.workload == 'c3':
        dims = [2 * D] + list(cfg['mlp_hidden_size'])
        fwd_flop_row = 8 * sum(a * b for a, b in zip(dims[:-1], dims[1:]))          # 4 products per cross unit, 2 flop per MAC
        flops = 3.0 * fwd_flop_row * rows_per_step                                   # forward + data gradient + weight gradient
        tf = flops / step_s / 1e12
        roof = {'bound': 'fp32', 'achieved': tf, 'peak': FP32_MFMA_PEAK_TFLOPS, 'unit': 'TFLOP/s', 'frac': tf / FP32_MFMA_PEAK_TFLOPS,
                'what': 'whole step (gather + towers fwd/bwd + optimizer) against the fp32 MFMA peak: %.1f kFLOP per row forward x 3 x %d rows'
                        % (fwd_flop_row / 1e3, rows_per_step), 'algorithmic_flops': flops, 'traffic': None}
        if world == 1 and getattr(model, 'fused_towers', False):
            # the three tower kernels alone: HIP events recorded in the library on the launch stream, a few eager steps after the timed region
            from recbole_cdr_amd import binding as B_
            B_.timing_enable(dev, 256)
            for i in range(8):
                b_ = batches[i % 4]
                opt.zero_grad(set_to_none=True)
                model.calculate_loss(b_).backward()
                opt.step()
            torch.cuda.synchronize()
            kt = {}
            for nm, ms in B_.timing_collect(dev):
                kt.setdefault(nm, []).append(ms)
            B_.timing_enable(dev, 0)
            ks, tot = [], 0.0
            # (a training step's forward also runs the data backward: conet_fb_kernel carries two of the three passes)
            for nm, share in (('conet_fb_kernel', 2.0), ('conet_fwd_kernel', 1.0), ('conet_bwd_kernel', 1.0), ('conet_wgrad_kernel', 1.0)):
                v = kt.get(nm, [])[2:]
                if v:
                    ms = sum(v) / len(v)
                    tot += ms
                    ks.append({'kernel': nm, 'avg_ms': ms, 'algorithmic_flops': fwd_flop_row * rows_per_step * share,
                               'achieved_TFLOPs': fwd_flop_row * rows_per_step * share / (ms * 1e-3) / 1e12})
            if tot > 0:
                roof['tower_kernels'] = {'sum_avg_ms': tot, 'achieved': flops / (tot * 1e-3) / 1e12, 'unit': 'TFLOP/s',
                                         'frac': flops / (tot * 1e-3) / 1e12 / FP32_MFMA_PEAK_TFLOPS}
            result['kernels'] = ks
    elif args.workload == 'c4':
        L = cfg['n_layers']
        nnz = 2 * (len(ds.s_pairs) + len(ds.t_pairs))                               # symmetric adjacency of both domains
        sparse_last = rowshard is None and bool(getattr(model, 'sparse_last_layer', False))
        full_layers = L - 1 if sparse_last else L                                  # the last layer runs on the batch's rows only
        spmm = nnz * (4 * D + 12) + 2 * (nu + ni) * 4 * D                           # one graph layer, one direction, both domains
        byts = 2.0 * full_layers * spmm + 7.0 * 4 * 2 * (nu + ni) * D              # full SpMMs fwd + bwd, + dense Adam
        gbs = byts / step_s / 1e9
        foot = 4.0 * 2 * (nu + ni) * D * 4 + nnz * 12.0                           # 4 tables + their Adam moments' share the SpMM touches + the adjacency
        gather_bw, stream_bw = measured_gather_bandwidth(dev, foot, D)
        roof_bw, roof_rec = cache_roof(foot, 4 * D)
        roof_bw = roof_bw or max(gather_bw, stream_bw)
        roof = {'bound': 'hbm', 'achieved': gbs, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': gbs / HBM_PEAK_GBS, 'algorithmic_bytes': byts,
                'frac_of_measured_cache_gather_rate': gbs / roof_bw, 'measured_cache_gather_rate_GBs': roof_bw,
                'peak_what': 'peak = the guide\'s HBM figure (MI355X_MICROARCH.md, 8 TB/s): the only roof the guide supports.  INFORMATIONAL beside it '
                             '(frac_of_measured_cache_gather_rate): the working set of this configuration (tables, gradients, adjacency: %.0f MB) sits in L2 / Infinity Cache, where the HBM peak '
                             'bounds nothing.  peak = the MEASURED rate of uniformly random %d-B row gathers (8 rows in flight per lane group) out of a '
                             'working set of that size: tools/mb_cache_bw.hip, filed as profiles/r04_mb_cache_bw.txt (builder run; 25 TB/s out of L2-resident '
                             '4 MB, 7.5 TB/s out of 77 MB, 5.8 TB/s out of HBM).  The product\'s own gather kernel (cdr_embloss_fwd) measured IN THIS RUN on the '
                             'same footprint: %.0f GB/s' % (foot / 1e6, 4 * D, gather_bw),
                'cache_roof_record': roof_rec, 'product_gather_kernel_GBs_measured_in_run': gather_bw, 'frac_of_hbm_peak_nominal': gbs / HBM_PEAK_GBS,
                'what': 'SpMM (4D + 12 B per nnz + 4D per output row, fwd + bwd, both domains) of the %d of %d layers that are evaluated on every row '
                        '+ dense Adam (7 x 4 B per table element)%s; the 43 MB of tables and the adjacency live in L2 / Infinity Cache, where the '
                        'SpMM gathers run at ~10 TB/s: NOMINAL fraction of the HBM peak, DESIGN 4.10'
                        % (full_layers, L, '; the last layer, restricted to the rows the loss gathers, is NOT counted (lower bound on the bytes moved)'
                           if sparse_last else ''),
                'reference_formulation_bytes': 2.0 * L * spmm + 7.0 * 4 * 2 * (nu + ni) * D, 'traffic': None}
    else:
        per_row = (3 * 4 * D + 24) if pairwise else (2 * 4 * D + 20)
        tabs_el = sum(p.numel() for p in model.parameters() if p.grad is not None)
        byts = float(rows_per_step * per_row + 7 * 4 * tabs_el)
        gbs = byts / step_s / 1e9
        foot = 7.0 * 4 * tabs_el
        gather_bw, stream_bw = measured_gather_bandwidth(dev, foot, D)
        roof_bw, roof_rec = cache_roof(foot, 4 * D)
        roof_bw = roof_bw or max(gather_bw, stream_bw)
        n_launch = 4
        roof = {'bound': 'hbm', 'achieved': gbs, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': gbs / HBM_PEAK_GBS, 'algorithmic_bytes': byts,
                'frac_of_measured_cache_gather_rate': gbs / roof_bw, 'measured_cache_gather_rate_GBs': roof_bw, 'limited_by': 'launch latency',
                'peak_what': 'peak = the guide\'s HBM figure (8 TB/s).  INFORMATIONAL (measured_cache_gather_rate_GBs): the MEASURED rate of uniformly random %d-B row gathers out of a %.0f MB working set (L2-resident; tools/mb_cache_bw.hip, '
                             'profiles/r04_mb_cache_bw.txt, builder run).  The step is not bound by it: it is %d dependent launches of ~5-10 us each '
                             '(producer, forward, backward, Adam) -- ms_per_step and launches_per_step are the figures that matter here' % (4 * D, foot / 1e6, n_launch),
                'cache_roof_record': roof_rec, 'product_gather_kernel_GBs_measured_in_run': gather_bw, 'launches_per_step': n_launch,
                'frac_of_hbm_peak_nominal': gbs / HBM_PEAK_GBS,
                'what': 'gather (%d B per row) + dense Adam over the parameters that received a gradient (7 x 4 B per element)' % per_row, 'traffic': None}
    result['roofline'] = roof
    if rowshard is not None:
        p_ = rowshard.part
        D_, L_ = cfg['embedding_size'], cfg['n_layers']
        layer_bytes = float((world - 1) * p_.nl * 4 * D_ * 4 * L_)                  # 2 domains x (L forward + L backward) all-gathers of [nl, D]
        if rowshard.batch_loss == 'replicated':
            result['exchange'] = {'all_gather_bytes_received_per_rank_per_step': layer_bytes + float((world - 1) * p_.nl * 4 * D_ * 2 * (L_ + 1)),
                                  'collectives_per_step': 2 * (2 * L_ + 1), 'batch_loss': 'replicated',
                                  'note': 'per domain: L all-gathers of E forward, L of g(1+E) backward, one of the stacked outputs'}
        else:
            Wx = (L_ + 1) * D_ if cfg['connect_way'] == 'concat' else 2 * D_
            Bs = -(-(rows_per_step // 2) // world)
            row_bytes = float(2 * 2 * (world - 1) * 2 * Bs * Wx * 4)                  # 2 domains x (reduce-scatter + all-gather) of [2 Bs, Wx] per rank
            result['exchange'] = {'all_gather_bytes_received_per_rank_per_step': layer_bytes, 'batch_row_bytes_received_per_rank_per_step': row_bytes,
                                  'collectives_per_step': 2 * (2 * L_) + 2 * 2 + 1, 'batch_loss': 'routed',
                                  'note': 'per domain: L all-gathers of E forward, L of g(1+E) backward; the batch: one reduce-scatter of the owned rows of every '
                                          'batch row [N, 2 B/N, W] -> each rank scores its B/N slice, one all-reduce of six sums, one all-gather of the gradient rows '
                                          '(instead of the all-gather of the stacked [n, (L+1) D] tables: %.1f MB per rank and step at N = %d)'
                                          % ((world - 1) * p_.nl * 4 * D_ * 2 * (L_ + 1) / 1e6, world)}
    if rank == 0 and not args.no_cpu_baseline:
        result['cpu_baseline'] = cpu_baseline_model(args, ds, cfg, S, k, batches[0] if pairwise else None)
    return result


def cpu_baseline_model(args, ds, cfg, S, k, pair_batch=None):
    """The oracle's calculate_loss + autograd + torch.optim.Adam (the reference's literal loop) on the host cores."""
    import numpy as np
    from oracle import conet as oconet, bitgcf as obit
    from oracle.common import IdSpace
    ids = IdSpace(ds.num_overlap_user, ds.num_target_only_user, ds.num_source_only_user, ds.num_overlap_item,
                  ds.num_target_only_item, ds.num_source_only_item)
    torch.manual_seed(0)
    D = cfg['embedding_size'] if 'embedding_size' in cfg else cfg['source_embedding_size']
    nu, ni = ids.total_num_users, ids.total_num_items
    xav = lambda r, c: (torch.randn(r, c) * (2.0 / (r + c)) ** 0.5).requires_grad_(True)
    params = {f'{d}_{w}_embedding.weight': xav(nu if w == 'user' else ni, D) for d in ('source', 'target') for w in ('user', 'item')}
    graph = None
    if args.workload == 'c1':
        from oracle import cmf as ocmf
        params = {'user_embedding.weight': xav(nu, D), 'item_embedding.weight': xav(ni, D)}
        loss_fn = lambda b: ocmf.calculate_loss(params, ids, b, cfg['alpha'], cfg['lambda'], cfg['gamma'])
    elif args.workload == 'c2':
        from oracle import emcdr as oem
        for l, (a, b) in enumerate(((D, 128), (128, D))):
            params[f'mapping.{2 * l}.weight'] = xav(b, a)
            params[f'mapping.{2 * l}.bias'] = torch.zeros(b, requires_grad=True)
        loss_fn = lambda b: oem.calculate_loss(params, ids, b, 'SOURCE', 'BPR', cfg['reg_weight'])
    elif args.workload == 'c3':
        dims = [2 * D] + cfg['mlp_hidden_size']
        for l, (a, b) in enumerate(zip(dims[:-1], dims[1:])):
            for t in ('source', 'target'):
                params[f'{t}_crossunit_linear.{l}.weight'] = xav(b, a)
                params[f'{t}_crossunit_linear.{l}.bias'] = torch.zeros(b, requires_grad=True)
            params[f'crossparas.{l}.weight'] = xav(b, a)
        for t in ('source', 'target'):
            params[f'{t}_outputunit.0.weight'] = xav(1, dims[-1])
            params[f'{t}_outputunit.0.bias'] = torch.zeros(1, requires_grad=True)
        loss_fn = lambda b: oconet.calculate_loss(params, ids, b)
    else:
        graph = obit.build_graph(ds.s_pairs, ds.t_pairs, nu, ni)
        loss_fn = lambda b: sum(obit.calculate_loss(params, ids, graph, b, cfg['n_layers'], cfg['lambda_source'],
                                                    cfg['lambda_target'], cfg['connect_way'], cfg['reg_weight']))
    opt = torch.optim.Adam(list(params.values()), lr=1e-3)
    rng = np.random.RandomState(1)
    if pair_batch is not None:
        batch = ds.pairwise_batch('source', S, k, rng, 'cpu')
    else:
        batch = dict(ds.pointwise_batch('source', S, k, rng, 'cpu'), **ds.pointwise_batch('target', S, k, rng, 'cpu'))
    ncores = host_cores()
    best = (None, 0.0)
    def step():
        opt.zero_grad()
        loss_fn(batch).sum().backward()
        opt.step()
    for nt in sorted({min(ncores, t) for t in (4, 8, 16, 32, 64)}):
        torch.set_num_threads(nt)
        step()
        t0 = time.perf_counter(); step(); rate = 1.0 / (time.perf_counter() - t0)
        if rate > best[1]:
            best = (nt, rate)
    rows = S * k if pair_batch is not None else 2 * S * (1 + k)
    def measure(nt, seconds):
        torch.set_num_threads(nt)
        step()
        t0 = time.perf_counter(); n = 0
        while (time.perf_counter() - t0 < seconds or n < 5) and n < 100:          # at least 5 steps whatever the time box
            step(); n += 1
        return rows * n / (time.perf_counter() - t0), n
    rate, n = measure(best[0], args.cpu_seconds)
    rate_all, n_all = measure(ncores, min(args.cpu_seconds, 5.0))
    rate_one, n_one = measure(1, min(args.cpu_seconds, 5.0))
    return {'value': rate, 'unit': 'interactions/s', 'cores': best[0], 'host_cores': ncores, 'os_cpu_count': os.cpu_count(), 'kind': 'port',
            'sample': '%d steps of %d rows, oracle calculate_loss + autograd + dense torch.optim.Adam, same table sizes and batch shape as the '
                      'GPU line, %d threads (best of a sweep)' % (n, rows, best[0]),
            'all_cores': {'value': rate_all, 'unit': 'interactions/s', 'cores': ncores, 'sample': '%d steps, same shape' % n_all},
            'one_thread': {'value': rate_one, 'unit': 'interactions/s', 'cores': 1, 'sample': '%d steps, same shape' % n_one}}


# ------------------------------------------------------------------------------------------------------ end-to-end leg
def e2e_leg(args, dev):
    """The whole path through the product's own loop at the headline's table sizes: EMCDR-BPR (D = args.dim) on args.users x
    2 x args.items_per_domain, synthetic interactions generated and kept on the device -> four-state loader with the device negative
    sampler -> CrossDomainTrainer(optimizer_mode='rowwise').fit over SOURCE, TARGET and OVERLAP (two epochs each: the first allocates
    the row-wise Adam state, the second is reported) -> Trainer.evaluate (fused mask + top-10) on 4,096 users.  Wall-clock rows/s per
    epoch INCLUDING shuffle, sampling, batching, Python and the loss read-back, next to the step-only rate of the same step objects on
    one resident batch of the same size (recbole_cdr/trainer/trainer.py:43-76, data/dataloader.py:114-162,
    sampler/crossdomain_sampler.py:139-175)."""
    import numpy as np
    from recbole_cdr_amd.data import CrossDomainDataloader, OverlapDataloader, DomainTrainLoader, FullSortEvalLoader
    from recbole_cdr_amd.data.synthetic import DeviceSyntheticDataset
    from recbole_cdr_amd.model.cross_domain_recommender.emcdr import EMCDR
    from recbole_cdr_amd.sampler import DeviceNegSampler
    from recbole_cdr_amd.trainer import CrossDomainTrainer
    from recbole_cdr_amd.utils import InputType
    OU, NI, B = int(args.users), int(args.items_per_domain), int(args.batch)
    steps_per_epoch = 8
    t_setup = time.perf_counter()
    ds = DeviceSyntheticDataset(OU=OU, TOU=0, SOU=0, OI=1, TOI=NI, SOI=NI, n_source_inter=steps_per_epoch * B + B // 2,
                                n_target_inter=steps_per_epoch * B + B // 2 + 20000, device=dev)
    cfg = {'source_domain': {'NEG_PREFIX': 'neg_'}, 'target_domain': {'NEG_PREFIX': 'neg_'}, 'device': dev, 'latent_factor_model': 'BPR',
           'source_embedding_size': args.dim, 'target_embedding_size': args.dim, 'reg_weight': 0.01, 'mapping_function': 'linear',
           'mlp_hidden_size': [128], 'learning_rate': 1e-3, 'optimizer_mode': 'rowwise', 'train_modes': ['SOURCE', 'TARGET', 'OVERLAP'],
           'epoch_num': ['2', '2', '2'], 'source_split': False, 'eval_step': 0, 'epochs': 2, 'topk': [10], 'valid_metric': 'Recall@10',
           'graph_step': os.environ.get('CDR_E2E_GRAPH', '1') != '0'}
    torch.manual_seed(2022)
    with torch.device(dev):
        model = EMCDR(cfg, ds)                                   # 4 tables created (and xavier-initialised) on the device
    held = 20000
    t_tr, t_te = ds.t_pairs[held:], ds.t_pairs[:held]
    gen = torch.Generator(device=dev); gen.manual_seed(7)
    OB = B
    train = CrossDomainDataloader(
        DomainTrainLoader({'source_user_id': ds.s_pairs[:, 0].contiguous(), 'source_item_id': ds.s_pairs[:, 1].contiguous()}, 'source_user_id',
                          'source_item_id', 'source_label', 'neg_', B, 1, InputType.PAIRWISE, DeviceNegSampler(ds, 'source', ds.s_pairs, dev),
                          shuffle=True, generator=gen),
        DomainTrainLoader({'target_user_id': t_tr[:, 0].contiguous(), 'target_item_id': t_tr[:, 1].contiguous()}, 'target_user_id',
                          'target_item_id', 'target_label', 'neg_', B, 1, InputType.PAIRWISE, DeviceNegSampler(ds, 'target', ds.t_pairs, dev),
                          shuffle=True, generator=gen),
        OverlapDataloader(OU, OB, device=dev, shuffle=True, generator=gen))
    torch.cuda.synchronize()
    t_setup = time.perf_counter() - t_setup
    trainer = CrossDomainTrainer(cfg, model)
    orig, log = trainer._train_epoch, []

    def timed(data, e):
        torch.cuda.synchronize(); t = time.perf_counter()
        v = orig(data, e)
        torch.cuda.synchronize(); log.append((time.perf_counter() - t, v))
        return v
    trainer._train_epoch = timed
    trainer.fit(train)
    rows = {'SOURCE': len(ds.s_pairs), 'TARGET': len(t_tr), 'OVERLAP': OU}
    out = {'what': 'EMCDR-BPR D=%d, %d users x 2 x %d items, B=%d, OB=%d, CrossDomainTrainer.fit(optimizer_mode=rowwise) on device loaders; '
                   'second epoch of each phase (the first allocates the row-wise Adam state)' % (args.dim, OU, NI, B, OB),
           'via': 'CrossDomainTrainer.fit', 'setup_s': t_setup, 'phases': {}}
    for j, ph in enumerate(('SOURCE', 'TARGET', 'OVERLAP')):
        sec, loss = log[2 * j + 1]
        out['phases'][ph] = {'epoch_ms': sec * 1e3, 'rows': rows[ph], 'rows_per_s': rows[ph] / sec, 'first_epoch_ms': log[2 * j][0] * 1e3,
                             'epoch_loss_sum': loss}
    with torch.no_grad():                                         # (exact fp64 sums of the state fit() trained: same digits in every invocation)
        out['state_checksum_after_fit'] = {n_: repr(exact_sum(p_)) for n_, p_ in model.named_parameters()}
    # ---- the SOURCE and the TARGET epoch side by side on two HIP streams (config['parallel_domains'] on one GPU) ----------------------
    cfg2 = dict(cfg, parallel_domains=True, train_modes=['SOURCE', 'TARGET'], epoch_num=['1', '1'])
    tr2 = CrossDomainTrainer(cfg2, model)
    tr2.fit(train)                                                # (creates the streams and their native contexts)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    tr2.fit(train)
    torch.cuda.synchronize(); sec2 = time.perf_counter() - t0
    seq = out['phases']['SOURCE']['epoch_ms'] + out['phases']['TARGET']['epoch_ms']
    with torch.no_grad():
        out['state_checksum_after_two_stream_epochs'] = {n_: repr(exact_sum(p_)) for n_, p_ in model.named_parameters()}
    out['source_and_target_on_two_streams'] = {'ms': sec2 * 1e3, 'rows': rows['SOURCE'] + rows['TARGET'],
                                               'rows_per_s': (rows['SOURCE'] + rows['TARGET']) / sec2, 'sequential_ms': seq,
                                               'speedup_over_sequential_phases': seq / (sec2 * 1e3),
                                               'what': 'one SOURCE epoch and one TARGET epoch enqueued side by side (CrossDomainTrainer with parallel_domains: '
                                                       'disjoint tables and optimizer state, bit-identical to the sequential phases)'}
    # ---- step-only rate of the same step objects on one resident full batch -----------------------------------------------------
    from recbole_cdr_amd.utils import train_mode2state
    for ph in ('SOURCE', 'TARGET', 'OVERLAP'):
        train.set_mode(train_mode2state[ph])
        model.set_phase(ph)
        prod = train.device_producer()
        it = iter(train)
        prod.resync()
        prod.launch()
        n = OB if ph == 'OVERLAP' else B
        for _ in range(2):
            model.fused_train_step(prod.fields, lr=1e-3)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(8):
            model.fused_train_step(prod.fields, lr=1e-3)
        torch.cuda.synchronize()
        step_rate = 8 * n / (time.perf_counter() - t0)
        out['phases'][ph]['step_only_rows_per_s'] = step_rate
        out['phases'][ph]['epoch_over_step_only'] = out['phases'][ph]['rows_per_s'] / step_rate
        for dl in (train.source_dataloader, train.target_dataloader, train.overlap_dataloader):
            dl.pr = 0
    # ---- evaluation: fused mask + top-10 over the whole catalogue ----------------------------------------------------------------
    model.set_phase('OVERLAP')
    te_users = torch.unique(t_te[:, 0])[:4096]
    keep = torch.isin(t_te[:, 0], te_users)
    hist = t_tr[torch.isin(t_tr[:, 0], te_users)]
    loader = FullSortEvalLoader('target_user_id', t_te[keep], hist, ds.num_overlap_item + ds.num_target_only_item, 0, dev, users_per_batch=1024)
    res = trainer.evaluate(loader)                                # first call sizes the workspaces
    torch.cuda.synchronize(); t0 = time.perf_counter()
    res = trainer.evaluate(loader)
    torch.cuda.synchronize(); sec = time.perf_counter() - t0
    nU, N = int(te_users.numel()), ds.num_overlap_item + ds.num_target_only_item
    out['evaluate'] = {'users': nU, 'items': N, 'ms': sec * 1e3, 'items_per_s': nU * N / sec, 'recall@10': res['recall@10'],
                       'what': 'Trainer.evaluate: mapped-user rows x the target item table, history + PAD masked, top-10, metrics'}
    return out


# ------------------------------------------------------------------------------------------------------ ingest leg
def synthetic_tokens(n_ids, lo, dev, gen, prefix=b'u'):
    """Packed tokens (bytes uint8, offsets int64 [n + 1]) of ``prefix`` + the decimal digits of a random permutation of [lo, lo + n_ids),
    built with torch ops on the device (setup, not timed): mixed lengths, so byte order differs from numeric order ('u10' < 'u2')."""
    ids = torch.randperm(n_ids, device=dev, generator=gen) + lo
    p10 = torch.tensor([10 ** k for k in range(9, -1, -1)], device=dev, dtype=torch.int64)         # up to 10 digits
    nd = torch.ones_like(ids)
    for k in range(1, 10):
        nd += (ids >= 10 ** k).to(torch.int64)
    L = len(prefix)
    lens = nd + L
    off = torch.zeros(n_ids + 1, device=dev, dtype=torch.int64)
    torch.cumsum(lens, 0, out=off[1:])
    out = torch.empty(int(off[-1]), device=dev, dtype=torch.uint8)
    step = 1 << 24                                                          # (a [n, 10] digit matrix of 50 M ids in slices)
    for a in range(0, n_ids, step):
        b = min(a + step, n_ids)
        d = ((ids[a:b, None] // p10[None, :]) % 10 + 48).to(torch.uint8)                            # right-aligned digits
        keep = torch.arange(10, device=dev)[None, :] >= (10 - nd[a:b, None])
        body = d[keep]                                                                              # row-major: each id's digits in order
        pos = off[a:b, None] + L + (torch.arange(10, device=dev)[None, :] - (10 - nd[a:b, None]))
        out[pos[keep]] = body
        for j, ch in enumerate(prefix):
            out[off[a:b] + j] = ch
    return out, off


def ingest_leg(args, dev):
    """The overlap id remap of the C5 id space on the device (csrc/cdr_remap_dev.hip; dataset.py:344-445 + :109-123): the USER field =
    `--users` tokens per domain, every user in both domains (all overlapped, each domain in its own shuffled order); the ITEM field =
    `--items-per-domain` tokens per domain, disjoint.  Token bytes and offsets are resident in HBM when the timed region starts, ids
    come back in HBM.  Beside it on the host cores: the oracle (Python sets + sorted) and the library's single-threaded host form
    (cdr_overlap_remap) on bounded samples of the same token shape."""
    from recbole_cdr_amd.data import overlap_remap_packed, overlap_remap
    from recbole_cdr_amd.data.remap import _pack
    from recbole_cdr_amd import binding as B_
    import ctypes
    import numpy as np
    gen = torch.Generator(device=dev).manual_seed(2022)
    nU, nI = int(args.users) - 1, int(args.items_per_domain)
    out = {'what': 'CrossDomainDataset overlap remap (dataset.py:344-445, :109-123) of one field of both domains: all occurrences radix-sorted '
                   'together in byte order, runs = distinct tokens, ids from class-wise scans; bit-exact with the host form '
                   '(tests/test_gpu_remap.py)', 'fields': {}}
    for name, (ns, nt, lo_s, lo_t) in {'users': (nU, nU, 1, 1), 'items': (nI, nI, 1, 1 + nI)}.items():
        sb, so = synthetic_tokens(ns, lo_s, dev, gen, b'u' if name == 'users' else b'i')
        tb, to = synthetic_tokens(nt, lo_t, dev, gen, b'u' if name == 'users' else b'i')
        torch.cuda.synchronize()
        overlap_remap_packed((sb[:1000], so[:101], None), (tb[:1000], to[:101], None), dev)         # warm-up: module load
        torch.cuda.synchronize()
        best, passes, counts = None, 0, None
        for _ in range(3):
            t0 = time.perf_counter()
            sid, tid, counts, passes = overlap_remap_packed((sb, so, None), (tb, to, None), dev)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            best = dt if best is None or dt < best else best
        c = counts.tolist()
        ok = (c == [nU + 1, 0, 0, nU + 1]) if name == 'users' else (c == [1, nI, nI, 2 * nI + 1])
        # property at full size: ids are a bijection onto their range, and byte order of tokens == id order inside a class
        chk = bool(torch.equal(torch.sort(sid).values, torch.arange(1, ns + 1, device=dev) + (0 if name == 'users' else nI))) if ok else False
        out['fields'][name] = {'source_tokens': ns, 'target_tokens': nt, 'token_bytes': int(sb.numel() + tb.numel()), 'seconds': best,
                               'tokens_per_s': (ns + nt) / best, 'radix_passes': passes, 'counts4': c, 'counts_ok': bool(ok), 'ids_are_a_permutation': chk}
        del sb, so, tb, to, sid, tid
        torch.cuda.empty_cache()
    tot = sum(f['source_tokens'] + f['target_tokens'] for f in out['fields'].values())
    sec = sum(f['seconds'] for f in out['fields'].values())
    out['value'], out['unit'], out['seconds'] = tot / sec, 'tokens/s', sec
    if not args.no_cpu_baseline:
        # host baselines on bounded samples of the same token shape (users: all overlapped, shuffled)
        rng = np.random.RandomState(2022)
        def sample(n):
            a, b = rng.permutation(n) + 1, rng.permutation(n) + 1
            return np.char.add('u', a.astype(str)).tolist(), np.char.add('u', b.astype(str)).tolist()
        from oracle import remap as oremap
        n_or = 1_000_000
        s, t = sample(n_or)
        t0 = time.perf_counter()
        ms, _, mt, _, _ = oremap.overlap_remap(s, ['x'], t, ['y'])
        a_s, a_t = oremap.apply_remap(s, ms), oremap.apply_remap(t, mt)
        t_or = time.perf_counter() - t0
        n_h = 4_000_000
        s2, t2 = sample(n_h)
        sbb, soo, snn = _pack(s2); tbb, too, tnn = _pack(t2)
        sid = np.empty(n_h, np.int64); tid = np.empty(n_h, np.int64); c4 = np.zeros(4, np.int64)
        vp = lambda a: a.ctypes.data_as(ctypes.c_void_p)
        t0 = time.perf_counter()
        B_.call('cdr_overlap_remap', sbb, vp(soo), vp(snn), n_h, tbb, vp(too), vp(tnn), n_h, vp(sid), vp(tid), vp(c4))
        t_h = time.perf_counter() - t0
        # the same sample through the device form INCLUDING the host-to-device copy of bytes + offsets and the copy of the ids back
        u8 = lambda b: np.frombuffer(b, np.uint8)
        t0 = time.perf_counter()
        ds, dt_, _c, _p = overlap_remap_packed((u8(sbb), soo, None), (u8(tbb), too, None), dev)
        hs, ht = ds.cpu().numpy(), dt_.cpu().numpy()
        t_pcie = time.perf_counter() - t0
        same = bool(np.array_equal(hs, sid) and np.array_equal(ht, tid))
        out['cpu_baseline'] = {'value': 2 * n_or / t_or, 'unit': 'tokens/s', 'cores': 1, 'kind': 'port',
                               'sample': '%d + %d user tokens, oracle/remap.py (Python sets, sorted, dict lookups), one thread' % (n_or, n_or)}
        out['host_form'] = {'value': 2 * n_h / t_h, 'unit': 'tokens/s', 'cores': 1,
                            'sample': '%d + %d user tokens, cdr_overlap_remap (csrc/cdr_remap.cpp: hash sets + std::sort), one thread' % (n_h, n_h)}
        out['device_form_from_host_buffers'] = {'value': 2 * n_h / t_pcie, 'unit': 'tokens/s', 'bit_equal_to_host_form': same,
                                                'sample': '%d + %d tokens, H2D of bytes + offsets and D2H of the ids inside the timed region' % (n_h, n_h)}
        out['vs_cpu'] = out['value'] / out['cpu_baseline']['value']
        out['vs_host_form'] = out['value'] / out['host_form']['value']
    return out


# ------------------------------------------------------------------------------------------------------ CoNet full-sort leg
def conet_fullsort_leg(args, dev):
    """Metric 2 for BASELINE configs[2]: CoNet.full_sort_predict (conet.py:222-242: the target tower, no cross terms, over every
    item) through the product model -- gather of the user rows, the two halves of the separable first layer (P = items W1i^T once,
    Q = users W1u^T + b1), then cdr_conet_fullsort for all U x N pairs in one launch.  C3's catalogue (18,564 target items) and a
    1,000,001-item synthetic one, U = 1 (recbole's default eval batch of 4,096 // N users) and U = 64; the oracle's per-user loop on
    the host cores beside it (bounded sample)."""
    from recbole_cdr_amd.data.synthetic import SyntheticCrossDomainDataset
    from recbole_cdr_amd.model.cross_domain_recommender.conet import CoNet
    from recbole_cdr_amd import binding as B_
    D, layers = 128, [64, 32, 16, 8]
    pair_flop = 2.0 * sum(a * b for a, b in zip(layers[:-1], layers[1:])) + 2.0 * layers[-1]      # behind the hoisted first layer
    out = {'what': 'CoNet.full_sort_predict, D=%d, tower [%d,%s]; %.0f FLOP per (user, item) pair behind the separable first layer '
                   '(2 N D h1 more, once per call, for P)' % (D, 2 * D, ','.join(map(str, layers)), pair_flop), 'cases': {}}
    for N in (18564, 1_000_001):
        ds = SyntheticCrossDomainDataset(OU=5983, TOU=2000, SOU=2000, OI=1, TOI=N - 1, SOI=1000, n_source_inter=2000, n_target_inter=2000)
        cfg = {'source_domain': {'NEG_PREFIX': 'neg_'}, 'target_domain': {'NEG_PREFIX': 'neg_'}, 'device': dev, 'embedding_size': D,
               'reg_weight': 0.01, 'mlp_hidden_size': layers}
        torch.manual_seed(2022)
        model = CoNet(cfg, ds).to(dev)
        model.eval()
        model.freeze_for_eval()                                # what CrossDomainTrainer.evaluate does around its loop of full_sort_predict calls
        for Uu in (1, 64):
            inter = {model.TARGET_USER_ID: torch.arange(1, 1 + Uu, device=dev, dtype=torch.int64)}
            for _ in range(3):
                sc = model.full_sort_predict(inter)
            reps = 20 if N < 100000 else 5
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                sc = model.full_sort_predict(inter)
            e1.record(); torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / reps
            # the one-launch kernel alone (HIP events around it on the launch stream)
            from recbole_cdr_amd import functional as F_
            ue = F_.gather_rows(model.target_user_embedding.weight, inter[model.TARGET_USER_ID])
            W1 = model.target_crossunit_linear[0].weight
            P = F_.gemm(model.target_item_embedding.weight[:model.target_num_items], W1[:, D:], trans_b=True)
            Q = F_.gemm(ue, W1[:, :D], trans_b=True, bias=model.target_crossunit_linear[0].bias)
            tail = list(model.target_crossunit_linear)[1:]
            lo = model.target_outputunit[0]
            res = torch.empty(Uu, P.shape[0], device=dev)
            call = lambda: F_.conet_fullsort(P, Q, [l.weight for l in tail], [l.bias for l in tail], lo.weight, lo.bias, out=res)
            call(); torch.cuda.synchronize()
            e0.record()
            for _ in range(reps):
                call()
            e1.record(); torch.cuda.synchronize()
            kms = e0.elapsed_time(e1) / reps
            Nn = int(sc.shape[1])
            flops = pair_flop * Uu * Nn
            byts = 4.0 * (Nn * layers[0] + Uu * layers[0] + Uu * Nn)                       # P read once, Q, scores written
            case = {'items_per_s': Uu * Nn / (ms * 1e-3), 'ms': ms, 'U': Uu, 'N': Nn,
                    'path': 'one launch: cdr_conet_fullsort_users (Q formed inside; evaluation-mode call packed once)' if model.__dict__.get('_eval_few') is not None and model.__dict__['_eval_few'].takes(inter[model.TARGET_USER_ID])
                    else 'gather + Q contraction + cdr_conet_fullsort (P cached in evaluation mode)',
                    'kernel': {'name': 'conet_fullsort_kernel', 'avg_ms': kms, 'algorithmic_flops': flops, 'achieved_TFLOPs': flops / (kms * 1e-3) / 1e12,
                               'frac_fp32_mfma_peak': flops / (kms * 1e-3) / 1e12 / FP32_MFMA_PEAK_TFLOPS,
                               'algorithmic_bytes': byts, 'achieved_GBs': byts / (kms * 1e-3) / 1e9, 'frac_hbm_peak': byts / (kms * 1e-3) / 1e9 / HBM_PEAK_GBS},
                    'bound': 'mfma' if flops / (FP32_MFMA_PEAK_TFLOPS * 1e12) > byts / (HBM_PEAK_GBS * 1e9) else 'hbm'}
            out['cases']['N=%d U=%d' % (Nn, Uu)] = case
        if N == 18564 and not args.no_cpu_baseline:
            from oracle import conet as oconet
            from oracle.common import IdSpace
            ids = IdSpace(ds.num_overlap_user, ds.num_target_only_user, ds.num_source_only_user, ds.num_overlap_item,
                          ds.num_target_only_item, ds.num_source_only_item)
            params = {k: v.detach().cpu() for k, v in model.state_dict().items()}
            ncores = host_cores()
            torch.set_num_threads(ncores)
            inter_c = {'target_user_id': torch.arange(1, 9)}
            with torch.no_grad():
                oconet.full_sort_predict(params, ids, {'target_user_id': torch.arange(1, 3)})
                t0 = time.perf_counter(); n = 0
                while time.perf_counter() - t0 < min(args.cpu_seconds, 5.0) or n < 2:
                    oconet.full_sort_predict(params, ids, inter_c); n += 1
                dt = (time.perf_counter() - t0) / n
            out['cpu_baseline'] = {'value': 8 * N / dt, 'unit': 'items/s', 'cores': ncores, 'kind': 'port',
                                   'sample': '%d calls of the oracle\'s full_sort_predict (the reference\'s per-user loop), 8 users x %d items, torch CPU' % (n, N)}
        del model
        torch.cuda.empty_cache()
    return out


# ------------------------------------------------------------------------------------------------------ CPU baseline
def cpu_baseline(args):
    """The oracle's row-wise step (oracle/train_step.py: same loss, same per-row gradients, lazy Adam) timed on this
    node's host cores on a bounded sample: down-scaled tables (host RAM), same D, batches of 65,536 triples."""
    from oracle import train_step as ts
    ncores = host_cores()
    D = args.dim
    nu, ni, B = 2_000_000, 1_000_000, 65536
    g = torch.Generator(); g.manual_seed(2022)
    U = torch.empty(nu, D).normal_(0, 0.01, generator=g)
    I = torch.empty(ni, D).normal_(0, 0.01, generator=g)
    us, is_ = ts.RowwiseAdamState(U), ts.RowwiseAdamState(I)
    mk = lambda hi: torch.randint(1, hi, (B,), generator=g)
    # FIXED thread count = the host's cores (round 6: the round-5 "best of a sweep" figure moved 4.5 x between two boxes of one CPU model --
    # a single timed step per candidate decided it); `value` = the MEDIAN of >= 5 samples of >= 20 steps each.  The sweep over thread counts
    # is still taken (short), but only reported (`thread_sweep`, detail file).
    step_no = [0]
    def one_step():
        step_no[0] += 1
        ts.rowwise_step(U, I, us, is_, mk(nu), mk(ni), mk(ni), step_no[0], opt=args.opt)
    def sample(nt, steps):
        torch.set_num_threads(nt)
        one_step()                                  # (thread pool + first touch at this count outside the sample)
        t0 = time.perf_counter()
        for _ in range(steps):
            one_step()
        return B * steps / (time.perf_counter() - t0)
    used = ncores
    sweep = {}
    for nt in sorted({min(ncores, t) for t in (1, 4, 8, 16, 32, 64, ncores)}):
        t0 = time.perf_counter()
        sweep[str(nt)] = sample(nt, 3)
        if time.perf_counter() - t0 > 6.0 and nt != ncores:
            break
    sample(used, 5)                                 # warm at the count that is reported
    n_samples, per = 5, 20
    t_one = B / max(sweep.get(str(used)) or sample(used, 3), 1.0)
    while n_samples * per * t_one > max(args.cpu_seconds, 5.0) * 2.5 and per > 5:      # a slow host: shorter samples, still five of them
        per -= 5
    samples = sorted(sample(used, per) for _ in range(n_samples))
    rate, steps = samples[len(samples) // 2], n_samples * per
    rate_all, n_all = rate, steps
    rate_one, n_one = sweep.get('1', 0.0), 3
    # the reference's OWN formulation beside the port's: dense autograd (index_select backward = a dense [rows, D] gradient) and
    # torch.optim.Adam over every row of both tables each step (recbole Trainer + emcdr.py:110-131), same tables, same batch shape; bounded to a
    # few steps -- each one sweeps the 1.5 GB of tables several times
    ref_form = None
    try:
        from oracle import losses as ol
        torch.set_num_threads(ncores)
        Ud, Id = U.clone().requires_grad_(True), I.clone().requires_grad_(True)
        dopt = torch.optim.Adam([Ud, Id], lr=1e-3)
        def dense_step():
            u_, p_, n_ = mk(nu), mk(ni), mk(ni)
            dopt.zero_grad()
            ue, pe, ne = Ud[u_], Id[p_], Id[n_]
            (ol.bpr_loss((ue * pe).sum(1), (ue * ne).sum(1)) + 0.01 * ol.emb_loss(ue, pe)).sum().backward()
            dopt.step()
        dense_step()
        t0 = time.perf_counter(); nd = 0
        while (time.perf_counter() - t0 < min(args.cpu_seconds, 6.0) or nd < 2) and nd < 20:
            dense_step(); nd += 1
        ref_form = {'value': B * nd / (time.perf_counter() - t0), 'unit': 'interactions/s', 'cores': ncores,
                    'sample': '%d steps, the reference\'s formulation of the same step (autograd with dense table gradients + torch.optim.Adam over every '
                              'row), same tables and batches' % nd}
        del Ud, Id, dopt
    except Exception as e:                       # (a host without the RAM for the dense gradients and moments: the port's figure stands alone)
        ref_form = {'error': repr(e)[:200]}
    torch.set_num_threads(used)
    model = ''
    try:
        with open('/proc/cpuinfo') as f:
            model = [l.split(':', 1)[1].strip() for l in f if l.startswith('model name')][0]
    except Exception:
        pass
    sample_txt = ('EMCDR-BPR D=%d, oracle row-wise step (fwd+bwd+lazy %s), batches of %d triples on down-scaled tables %d users x %d items '
                  '(host RAM)' % (D, args.opt, B, nu, ni))
    return {'value': rate, 'unit': 'interactions/s', 'cores': used, 'host_cores': ncores, 'os_cpu_count': os.cpu_count(), 'kind': 'port', 'cpu_model': model,
            'sample': 'median of %d samples of %d steps, %s, torch %d threads (all host cores, fixed)' % (n_samples, per, sample_txt, used),
            'samples': samples, 'spread': (samples[-1] - samples[0]) / rate if rate else None,
            'thread_sweep': {'unit': 'interactions/s', 'what': '3 steps per thread count, reported only (value uses all host cores)', 'by_threads': sweep},
            'all_cores': {'value': rate_all, 'unit': 'interactions/s', 'cores': ncores, 'sample': '%d steps, same shape' % n_all},
            'one_thread': {'value': rate_one, 'unit': 'interactions/s', 'cores': 1, 'sample': '%d steps, same shape' % n_one},
            'reference_formulation': ref_form}


def run_replicas(args, world, rank, dev, attempts):
    """Last resort of ``--gpus N`` when no sharded layout came up (LayoutUnavailable): every rank runs the one-GPU headline step on its
    own GPU with its own full tables -- no data-path communication at all -- and the line reports N x the slowest rank's rate, with
    what failed (``layout_fallback``).  Weak scaling of independent replicas: says nothing about the collectives, and says so."""
    import copy
    import gc
    gc.collect(); torch.cuda.empty_cache()
    a = copy.copy(args)
    a.no_map = a.no_extra_legs = a.no_fullsort = a.no_config_legs = a.no_cpu_baseline = a.no_e2e = a.no_ingest = True
    a.force_shard = False
    r = run_c5(a, 1, 0, dev)
    ms = float(ctrl_max(torch.tensor([r['ms_per_step']], dtype=torch.float64)))
    r.update(value=world * 2 * args.batch / (ms * 1e-3), n_gpus=world, ms_per_step=ms, scaling='weak')
    r['config']['sharding'] = ('none: %d INDEPENDENT REPLICAS (fallback: no sharded layout came up within %.0f s on every rank) -- each rank '
                               'its own full tables and batches, no data-path communication' % (world, args.preflight_seconds))
    r['config']['sharding_mode'] = 'replicas'
    r['layout_fallback'] = {'used': 'replicas', 'attempts': attempts, 'fell_back': True}
    return r


def main():
    args = parse()
    if args.headline_only:
        args.no_map = args.no_extra_legs = args.no_fullsort = args.no_config_legs = args.no_cpu_baseline = args.no_e2e = args.no_ingest = True
    # the contract is ONE JSON line on stdout; RCCL and gloo print banners through C stdio (some only when the process exits),
    # so with a process group everything else written to fd 1 is sent to stderr and the line goes to the saved descriptor
    real_stdout = None
    if int(os.environ.get('WORLD_SIZE', '1')) > 1 or args.force_shard:
        sys.stdout.flush()
        real_stdout = os.dup(1)
        os.dup2(2, 1)
    world, rank, local = dist_setup(args)
    dev = torch.device('cuda', local)
    import recbole_cdr_amd  # noqa: F401  (raises loudly if libcdrhip.so is missing)
    if args.only_e2e:
        print(json.dumps({'e2e': e2e_leg(args, dev)}), flush=True)
        return
    if args.only_ingest:
        print(json.dumps({'ingest': ingest_leg(args, dev)}), flush=True)
        return
    if args.workload == 'c5' and (world > 1 or args.force_shard) and not args.single_layout:
        # N > 1: BOTH layouts of the C5 tables in one record -- north_star's row shard (rows r % N, row / gradient-row all-to-all)
        # and the dimension shard (D/N columns of every row, ids all-gathered, one partial score per triple all-reduced) -- so
        # that one hardware run shows them side by side.  The headline fields are those of --shard (default row: north_star's).
        import copy
        import gc
        try:
            result = run_c5(args, world, rank, dev)            # (brings its layout up under the preflight watchdog; args.shard = what came up)
        except LayoutUnavailable as e:
            result = run_replicas(args, world, rank, dev, e.attempts)
        if result['config'].get('sharding_mode') != 'replicas':
            first, second = args.shard, ('row' if args.shard == 'dim' else 'dim')
            gc.collect(); torch.cuda.empty_cache()
            a2 = copy.copy(args)
            a2.shard, a2.no_fullsort, a2.no_map, a2.no_extra_legs, a2.no_layout_fallback = second, True, True, True, True
            other = None
            try:
                other = run_c5(a2, world, rank, dev)
            except Exception as e:  # noqa: BLE001
                result.setdefault('leg_errors', {})['layout_' + second] = repr(e)[:500]
            pick = lambda r: None if r is None else {k: r.get(k) for k in ('value', 'unit', 'ms_per_step', 'scaling', 'n_gpus', 'exchange', 'kernels',
                                                                          'roofline', 'layout_fallback')} | {'sharding': r['config']['sharding']}
            result['layouts'] = {first: pick(result), second: pick(other)}
            # BASELINE.json's north_star names the ROW shard (item / user tables row-sharded, all-to-all of rows): since round 6 it is the
            # layout of the headline fields (--shard row is the default); the dimension layout (16 B per triple over xGMI instead of ~2 KB)
            # is timed into `layouts.dim` of the same record
            result['north_star_layout'] = 'row'
    elif args.workload == 'c5':
        try:
            result = run_c5(args, world, rank, dev)
        except LayoutUnavailable as e:
            result = run_replicas(args, world, rank, dev, e.attempts)
        if world == 1 and not args.no_config_legs:
            # BASELINE configs[0..3] behind the headline, compact: so that ONE default invocation covers all five configurations.
            # Each leg is the full `--workload cN` measurement at 200 steps (they take 0.03-1 ms each) with a 3-second CPU sample.
            import copy
            import gc
            gc.collect(); torch.cuda.empty_cache()
            legs = {}
            for wl, extra in (('c1', {}), ('c2', {}), ('c3', {}), ('c4', {'full_last_layer': True}), ('c4', {}),
                              ('c1', {'deterministic_leg': True}), ('c2', {'deterministic_leg': True})):
                a = copy.copy(args)
                a.workload, a.steps, a.warmup, a.cpu_seconds = wl, 200, 20, 3.0
                for k_, v_ in extra.items():
                    setattr(a, k_, v_)
                if wl == 'c4' and not extra:
                    a.no_cpu_baseline = True                 # same CPU formulation as the full-last-layer leg: measured once
                name = wl if not (wl == 'c4' and extra) else 'c4_full_last_layer'
                det_leg = bool(extra.get('deterministic_leg'))
                if det_leg:
                    # the same leg under functional.set_deterministic(True): the drop-in losses' dense gradients without float atomics
                    # (cdr_ordered_bwd at these batch sizes) -- what run-to-run reproducibility costs, in the driver's own run
                    from recbole_cdr_amd import functional as F_det
                    name, a.no_cpu_baseline = wl + '_deterministic', True
                    was_det = F_det.deterministic()
                    F_det.set_deterministic(True)
                try:
                    r = run_model_workload(a, world, rank, dev)
                    legs[name] = {k_: r.get(k_) for k_ in ('value', 'unit', 'ms_per_step', 'steps', 'roofline', 'kernels', 'cpu_baseline', 'final_loss')
                                  if r.get(k_) is not None}
                    legs[name]['workload'] = r['config']['workload']
                    legs[name]['rows_per_step'] = r['config']['rows_per_step']
                    cb = legs[name].get('cpu_baseline')
                    if cb:
                        legs[name]['vs_cpu'] = r['value'] / cb['value'] if cb.get('value') else None
                except Exception as e:  # noqa: BLE001
                    result.setdefault('leg_errors', {})['config_' + name] = repr(e)[:500]
                    print('bench: config leg %s failed: %r' % (name, e), file=sys.stderr)
                finally:
                    if det_leg:
                        F_det.set_deterministic(was_det)
                gc.collect(); torch.cuda.empty_cache()
            result['configs'] = legs
    else:
        result = run_model_workload(args, world, rank, dev)
    if world == 1 and rank == 0 and args.workload == 'c5' and not args.no_e2e:
        import gc
        gc.collect(); torch.cuda.empty_cache()
        try:
            result['e2e'] = e2e_leg(args, dev)
        except Exception as e:  # noqa: BLE001
            result.setdefault('leg_errors', {})['e2e'] = repr(e)[:500]
            print('bench: e2e leg failed: %r' % (e,), file=sys.stderr)
        gc.collect(); torch.cuda.empty_cache()
    if world == 1 and rank == 0 and args.workload == 'c5' and not args.no_ingest:
        import gc
        gc.collect(); torch.cuda.empty_cache()
        try:
            result['ingest'] = ingest_leg(args, dev)
        except Exception as e:  # noqa: BLE001
            result.setdefault('leg_errors', {})['ingest'] = repr(e)[:500]
            print('bench: ingest leg failed: %r' % (e,), file=sys.stderr)
        gc.collect(); torch.cuda.empty_cache()
    if world == 1 and rank == 0 and not args.no_fullsort and args.workload in ('c5', 'c3'):
        try:
            result.setdefault('fullsort', {})['conet'] = conet_fullsort_leg(args, dev)
        except Exception as e:  # noqa: BLE001
            result.setdefault('leg_errors', {})['fullsort_conet'] = repr(e)[:500]
            print('bench: conet fullsort leg failed: %r' % (e,), file=sys.stderr)
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline and args.workload == 'c5':
            result['cpu_baseline'] = cpu_baseline(args)
        from recbole_cdr_amd import functional as F_
        result['deterministic_backward'] = bool(F_.deterministic())      # CDR_DETERMINISTIC=1: the drop-in losses' dense gradients without float atomics
        if not result['deterministic_backward']:
            # which legs of THIS line are not bit-reproducible run to run: only the drop-in dense backward of the small configurations (fp32
            # atomics into dense gradients, as torch's own embedding backward); every O(batch) step (the headline, the sharded layouts, the
            # medium / k-major / pointwise steps, the OVERLAP step), the full-sort legs and the ingest sum in a fixed order
            result['nondeterministic_legs'] = ['configs.c1', 'configs.c2', 'configs.c4 (BiTGCF scatters)', 'e2e: none', 'headline: none']
            result['nondeterministic_legs'] = [x for x in result['nondeterministic_legs'] if not x.endswith(': none')]
        result['bench_wall_s'] = round(time.perf_counter() - T_START, 1)  # the whole invocation, imports and every leg included
        emit(result, real_stdout, detail_file=args.detail_file)
    if world > 1 or args.force_shard:
        import torch.distributed as dist
        stuck = any('still blocked' in str(e) for a in (result.get('layout_fallback') or {}).get('attempts', []) for e in a['errors'].values()) \
            if isinstance(result, dict) else False
        if stuck:
            sys.stderr.flush()
            os._exit(0)                           # a watchdog thread is still inside a communicator that never formed: do not wait for it
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
