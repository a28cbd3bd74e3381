"""world_size-2 CPU (gloo) test of the row-sharded training step's exchange logic (shard.ShardedBPRStep): routing by
owner, the four all-to-alls, the global loss reduction and the owner-side apply.  The arithmetic is injected from the
oracle (tests may use it); on the GPU the same class runs with libcdrhip (tests/test_gpu_parity.py::test_sharded_*)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


class OracleOps:
    """Stand-in compute for shard.ShardedBPRStep: the oracle's formulas on CPU tensors (same method set as NativeOps)."""

    def route(self, ids0, ids1, world):
        ids = ids0 if ids1 is None else torch.cat([ids0, ids1])
        owner = ids % world
        perm = torch.argsort(owner, stable=True).to(torch.int32)
        return perm, torch.bincount(owner, minlength=world)

    def permute(self, src0, src1, perm, divisor, flag_below=0):
        src = src0 if src1 is None else torch.cat([src0, src1])
        out = src[perm.long()] // divisor
        return out | ((perm.long() < flag_below).long() << 62)

    def inverse_perm(self, perm):
        pos = torch.empty(perm.numel(), dtype=torch.int64)
        pos[perm.long()] = torch.arange(perm.numel())
        return pos

    def gather_rows(self, table, local_ids):
        return table[local_ids].clone()

    def dedup(self, ids0, ids1, world, local_rows):
        ids = torch.cat([ids0, ids1])
        lb = max(int(local_rows - 1).bit_length(), 1)
        key = ((ids % world) << lb) | (ids // world)
        uniq, umap = torch.unique(key, return_inverse=True)                 # ascending key = grouped by owner, ascending local row
        return {'uniq_local': uniq & ((1 << lb) - 1), 'umap': umap, 'counts': torch.bincount(uniq >> lb, minlength=world), 'n': ids.numel()}

    def segsum(self, plan, G_rows, neg_start, rows, reg_coef, n_uniq):
        n = plan['n']
        sign = torch.cat([torch.ones(neg_start), -torch.ones(n - neg_start)])
        src = torch.cat([G_rows, G_rows[: n - neg_start]]) * sign[:, None]
        out = torch.zeros(n_uniq, G_rows.shape[1]).index_add_(0, plan['umap'], src)
        cnt = torch.zeros(n_uniq).index_add_(0, plan['umap'][:neg_start], torch.ones(neg_start))
        return out + float(reg_coef[0]) * cnt[:, None] * rows

    def fwd_grad(self, utab, itab, uidx, pidx, nidx, B_mean, gamma, reg_weight, out, GU, GI, scatter=True):
        u, p, n = utab[uidx], itab[pidx], itab[nidx]
        x = (u * p).sum(1) - (u * n).sum(1)
        s = torch.sigmoid(x)
        g = -(1.0 / B_mean) * (s * (1 - s)) / (gamma + s)
        B = uidx.numel()
        GU[:B] = g[:, None] * (p - n)
        if scatter:
            GI[pidx] = g[:, None] * u
            GI[nidx] = -g[:, None] * u
        else:
            GI[:B] = g[:, None] * u
        out[6] = (-torch.log(gamma + s)).sum()
        out[7] = (u * u).sum()
        out[8] = (p * p).sum()

    def batch_norms(self, utab, irows, u_loc, ip):
        u, p = utab[u_loc], irows[ip]
        return torch.stack([torch.zeros(()), (u * u).sum(), (p * p).sum()])

    def local_step(self, utab, ustate, irows, u_loc, ip, in_, B_mean, gamma, reg_weight, opt, hp, step, out):
        """shard.NativeOps.local_step in the oracle's arithmetic: forward + compact gradients, the user rows updated here (out[4] = the
        EmbLoss coefficient of the GLOBAL batch, set by the caller), GP returned for segsum; out[6:9] = this rank's sums."""
        Bl = u_loc.numel()
        GU, GP = torch.zeros(Bl, utab.shape[1]), torch.zeros(Bl, utab.shape[1])
        keep = out[4:6].clone()
        self.fwd_grad(utab, irows, u_loc, ip, in_, B_mean, gamma, reg_weight, out, GU, GP, scatter=False)
        out[4:6] = keep
        self.sort_apply(utab, ustate, u_loc, GU, opt, hp, step, reg_limit=Bl, reg_coef=out[4:5])
        return GP

    def finish_sums(self, sums3, B_mean, reg_weight, out):
        main = sums3[0] / B_mean
        nu, ni = sums3[1].sqrt(), sums3[2].sqrt()
        out[1], out[2], out[3] = main, nu, ni
        out[0] = main + reg_weight * (nu + ni) / B_mean
        out[4] = reg_weight / (B_mean * nu) if float(nu) > 0 else 0.0
        out[5] = reg_weight / (B_mean * ni) if float(ni) > 0 else 0.0

    def sort_apply(self, table, state, local_ids, grads, opt, hp, step, reg_limit=0, reg_coef=None, tagged=False):
        if local_ids.numel() == 0:
            return
        flags = ((local_ids >> 62) & 1).float() if tagged else (torch.arange(local_ids.numel()) < reg_limit).float()
        rows, inv = torch.unique(local_ids & ((1 << 62) - 1), return_inverse=True)
        Gs = torch.zeros(rows.numel(), table.shape[1]).index_add_(0, inv, grads)
        if reg_coef is not None:
            cnt = torch.zeros(rows.numel()).index_add_(0, inv, flags)
            Gs = Gs + reg_coef[0] * cnt[:, None] * table[rows]
        from oracle.train_step import _apply_rows, RowwiseAdamState
        if opt == 0:
            _apply_rows(table, None, rows, Gs, 'sgd', hp['lr'], step)
        else:
            st = RowwiseAdamState.__new__(RowwiseAdamState)
            st.m, st.v = state
            _apply_rows(table, st, rows, Gs, 'adam', hp['lr'], step, hp['b1'], hp['b2'], hp['eps'])

    # ---- round 6: the coarse ops of the direct form (shard.NativeOps.route_triples / plan / gather_rows_norms / norm_sums / shard_step / owner_apply)
    def route_triples(self, uid, pid, nid, world):
        owner = uid % world
        perm = torch.argsort(owner, stable=True)
        return torch.stack((uid[perm] // world, pid[perm], nid[perm]), 1).contiguous(), torch.bincount(owner, minlength=world)

    def plan(self, recv3, world, user_rows, item_local_rows, D, slot=0):
        d = self.dedup(recv3[:, 1].contiguous(), recv3[:, 2].contiguous(), world, item_local_rows)
        return {'Bl': recv3.shape[0], 'u_loc': recv3[:, 0].contiguous(), 'uniq_local': d['uniq_local'], 'umap': d['umap'], 'counts': d['counts'], 'n': d['n']}

    def gather_rows_norms(self, table, local_ids):
        rows = table[local_ids].clone()
        return rows, (rows * rows).sum(1)

    def norm_sums(self, utab, plan, nrm2, sums3):
        u = utab[plan['u_loc']]
        sums3[0], sums3[1], sums3[2] = 0.0, (u * u).sum(), nrm2[plan['umap'][:plan['Bl']]].sum()

    def shard_step(self, utab, ustate, irows, plan, n_uniq, B_mean, gamma, reg_weight, opt, hp, step, out):
        Bl = plan['Bl']
        keep = out[4:9].clone()
        GP = self.local_step(utab, ustate, irows, plan['u_loc'], plan['umap'][:Bl], plan['umap'][Bl:], B_mean, gamma, reg_weight, opt, hp, step, out)
        loss = out[6].clone()
        out[4:9] = keep                                          # the all-reduced norm sums (and the coefficients) stay; only out[6] is this call's
        out[6] = loss
        return self.segsum(plan, GP, Bl, irows, out[5:6], n_uniq)

    def owner_apply(self, table, state, ids, runs, grads, opt, hp, step):
        self.sort_apply(table, state, ids, grads, opt, hp, step)


def _free_port():
    s = socket.socket(); s.bind(('127.0.0.1', 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, opt, dedup, q, direct=True):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        import recbole_cdr_amd  # noqa: F401
        from recbole_cdr_amd.shard import ShardedBPRStep, shard_of
        torch.manual_seed(0)                                   # same full tables on every rank
        nu, ni, D, B, reg, lr = 41, 29, 8, 37, 0.03, 0.05
        U = torch.randn(nu, D) * 0.3
        I = torch.randn(ni, D) * 0.3
        Ul, Il = shard_of(U, world, rank), shard_of(I, world, rank)
        st = ShardedBPRStep(Ul, Il, nu, ni, B, opt=opt, lr=lr, reg_weight=reg, ops=OracleOps(), dedup=dedup, direct=direct)
        assert st.direct == (direct and dedup)
        losses = []
        batches = []
        for step in range(3):
            g = torch.Generator(); g.manual_seed(100 * step + rank)
            u = torch.randint(0, nu, (B,), generator=g); p = torch.randint(0, ni, (B,), generator=g)
            n = torch.randint(0, ni, (B,), generator=g)
            batches.append((u, p, n))
            out = st.step(u, p, n)
            losses.append(float(out[0]))
        # numpy (pickled by value): torch tensors would travel as shared-memory fds that die with the worker
        q.put((rank, Ul.numpy().copy(), Il.numpy().copy(), losses, [tuple(t.numpy().copy() for t in b) for b in batches]))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('dedup', [False, True, 'staged'])
@pytest.mark.parametrize('opt', ['sgd', 'adam'])
def test_sharded_step_matches_single_process(opt, dedup):
    direct = dedup is True                      # True: the round-6 direct form; 'staged': round 5's (dedup, per-item sums of every row); False: no dedup
    dedup = bool(dedup)
    from oracle import train_step as ts
    world = 2
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, opt, dedup, q, direct)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(world)], key=lambda t: t[0])
    res = [(r, torch.from_numpy(a), torch.from_numpy(b), l, [tuple(torch.from_numpy(x) for x in bb) for bb in bs])
           for r, a, b, l, bs in res]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # single-process reference on the concatenated global batch
    torch.manual_seed(0)
    nu, ni, D, reg, lr = 41, 29, 8, 0.03, 0.05
    U = torch.randn(nu, D) * 0.3
    I = torch.randn(ni, D) * 0.3
    us, is_ = ts.RowwiseAdamState(U), ts.RowwiseAdamState(I)
    for step in range(3):
        u = torch.cat([res[r][4][step][0] for r in range(world)])
        p = torch.cat([res[r][4][step][1] for r in range(world)])
        n = torch.cat([res[r][4][step][2] for r in range(world)])
        loss = ts.rowwise_step(U, I, us, is_, u, p, n, step + 1, opt=opt, lr=lr, reg_weight=reg)
        for r in range(world):
            assert abs(res[r][3][step] - float(loss)) <= 1e-5 * abs(float(loss)), (step, res[r][3][step], float(loss))
    for r in range(world):
        atol = lr * 1e-2 if opt == 'adam' else 1e-6
        torch.testing.assert_close(res[r][1], U[r::world], rtol=2e-5, atol=atol)
        torch.testing.assert_close(res[r][2], I[r::world], rtol=2e-5, atol=atol)


def test_shard_rows_partition():
    from recbole_cdr_amd.shard import shard_rows
    for total in (1, 7, 8, 50_000_001):
        for world in (1, 2, 4, 8):
            assert sum(shard_rows(total, world, r) for r in range(world)) == total
            for r in range(world):
                assert shard_rows(total, world, r) == len(range(r, total, world))


def _worker_pipe(rank, world, port, q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        import recbole_cdr_amd  # noqa: F401
        from recbole_cdr_amd.shard import ShardedBPRStep, shard_of, run_pipelined
        groups = [dist.new_group(list(range(world))) for _ in range(2)]
        torch.manual_seed(0)
        nu, ni, D, B = 53, 31, 8, 29
        tabs = [(torch.randn(nu, D) * 0.3, torch.randn(ni, D) * 0.3) for _ in range(2)]
        steps = [ShardedBPRStep(shard_of(U, world, rank), shard_of(I, world, rank), nu, ni, B, opt='sgd', lr=0.05,
                                reg_weight=0.02, group=groups[d], ops=OracleOps()) for d, (U, I) in enumerate(tabs)]
        out = []
        for it in range(2):
            bs = []
            for d in range(2):
                g = torch.Generator(); g.manual_seed(1000 * it + 10 * d + rank)
                bs.append((torch.randint(0, nu, (B,), generator=g), torch.randint(0, ni, (B,), generator=g),
                           torch.randint(0, ni, (B,), generator=g)))
            run_pipelined([steps[d].step_gen(*bs[d]) for d in range(2)])
            out.append([tuple(t.numpy().copy() for t in b) for b in bs])
        q.put((rank, [(st.U.numpy().copy(), st.I.numpy().copy(), float(st.out[0])) for st in steps], out))
    finally:
        dist.destroy_process_group()


def test_pipelined_two_domains_world3():
    """run_pipelined over two independent domain steps (own process group each), world_size 3, uneven buckets."""
    from oracle import train_step as ts
    world = 3
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_pipe, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=180) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    torch.manual_seed(0)
    nu, ni, D = 53, 31, 8
    tabs = [(torch.randn(nu, D) * 0.3, torch.randn(ni, D) * 0.3) for _ in range(2)]
    for d in range(2):
        U, I = tabs[d]
        loss = None
        for it in range(2):
            u = torch.cat([torch.from_numpy(res[r][2][it][d][0]) for r in range(world)])
            p = torch.cat([torch.from_numpy(res[r][2][it][d][1]) for r in range(world)])
            n = torch.cat([torch.from_numpy(res[r][2][it][d][2]) for r in range(world)])
            loss = ts.rowwise_step(U, I, None, None, u, p, n, it + 1, opt='sgd', lr=0.05, reg_weight=0.02)
        for r in range(world):
            Ur, Ir, lr_ = res[r][1][d]
            assert abs(lr_ - float(loss)) <= 1e-5 * abs(float(loss))
            torch.testing.assert_close(torch.from_numpy(Ur), U[r::world], rtol=2e-5, atol=1e-6)
            torch.testing.assert_close(torch.from_numpy(Ir), I[r::world], rtol=2e-5, atol=1e-6)


# ---------------------------------------------------------------------------------------------------------------------
# dp.ShardedDataParallel (flat parameter buffer, reduce-scatter + all-gather, sharded Adam) on CPU: a pure-torch stand-in
# model and injected Adam arithmetic, so the flattening / slicing / per-parameter step logic runs under gloo here; the
# native kernels run the same class in tests/test_gpu_parity.py::test_sharded_data_parallel_*.
class _TinyModel(torch.nn.Module):
    def __init__(self):
        super().__init__()
        g = torch.Generator(); g.manual_seed(0)
        self.a = torch.nn.Parameter(torch.randn(37, 5, generator=g))
        self.b = torch.nn.Parameter(torch.randn(11, generator=g))
        self.c = torch.nn.Parameter(torch.randn(23, 3, generator=g))       # gets no gradient in phase 0
        self.phase = 0

    def calculate_loss(self, x):
        loss = ((self.a[x['i']] * x['v']).sum(1) - self.b[x['j']]).pow(2).mean()
        if self.phase:
            loss = loss + self.c[x['k']].pow(2).sum()
        return loss


def _torch_adam_segments(sdp, segs):
    b1, b2 = sdp.betas
    for off, n, i in segs:
        sdp.steps[i] += 1
        t = int(sdp.steps[i])
        g = sdp.gshard[off:off + n]
        m, v, p = sdp.exp_avg[off:off + n], sdp.exp_avg_sq[off:off + n], sdp.pshard[off:off + n]
        m.mul_(b1).add_(g, alpha=1 - b1); v.mul_(b2).addcmul_(g, g, value=1 - b2)
        p.sub_((sdp.lr / (1 - b1 ** t)) * m / (v.sqrt() / (1 - b2 ** t) ** 0.5 + sdp.eps))


def _tiny_batch(step, rank):
    g = torch.Generator(); g.manual_seed(77 * step + rank)
    n = 9 + rank
    return {'i': torch.randint(0, 37, (n,), generator=g), 'j': torch.randint(0, 11, (n,), generator=g),
            'k': torch.randint(0, 23, (n,), generator=g), 'v': torch.randn(n, 5, generator=g)}


def _worker_dp(rank, world, port, q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        import recbole_cdr_amd  # noqa: F401
        from recbole_cdr_amd.dp import ShardedDataParallel
        model = _TinyModel()
        sdp = ShardedDataParallel(model, lr=0.05, adam_impl=_torch_adam_segments)
        for step in range(5):
            model.phase = int(step >= 2)
            sdp.step(_tiny_batch(step, rank))
        q.put((rank, {k: v.detach().numpy().copy() for k, v in model.named_parameters()}))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('world', [2, 3])
def test_sharded_data_parallel_gloo(world):
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_dp, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    ref = _TinyModel()
    opt = torch.optim.Adam(ref.parameters(), lr=0.05)
    for step in range(5):
        ref.phase = int(step >= 2)
        acc = {}
        for r in range(world):
            ref.zero_grad(set_to_none=True)
            ref.calculate_loss(_tiny_batch(step, r)).backward()
            for k, p in ref.named_parameters():
                if p.grad is not None:
                    acc[k] = acc.get(k, 0) + p.grad.clone() / world
        ref.zero_grad(set_to_none=True)
        for k, p in ref.named_parameters():
            if k in acc:
                p.grad = acc[k]
        opt.step()
    for r in range(world):
        for k, p in ref.named_parameters():
            torch.testing.assert_close(torch.from_numpy(res[r][1][k]), p.detach(), rtol=1e-5, atol=1e-6)


# ---------------------------------------------------------------------------------------------------------------------
# dimshard.DimShardedBPRStep (columns of every row on every rank; ids all-gathered, one partial score all-reduced) on CPU
class OracleDimOps:
    """Stand-in compute for dimshard.DimShardedBPRStep on CPU tensors: the formulas of csrc/cdr_dimshard.hip's two
    kernels, then the same per-table segment + apply as OracleOps.sort_apply on this rank's columns."""

    def __init__(self, user_cols, item_cols, opt, hp, gamma, reg_weight):
        self.U, self.I, self.opt, self.hp, self.gamma, self.reg_weight = user_cols, item_cols, opt, hp, gamma, reg_weight
        self.out = torch.zeros(12)
        self.state = [(torch.zeros_like(user_cols), torch.zeros_like(user_cols)), (torch.zeros_like(item_cols), torch.zeros_like(item_cols))]
        self.t = 0

    def pack_ids(self, uid, pid, nid, out32):
        out32.copy_(torch.cat([uid, pid, nid]).to(torch.int32))

    def unpack_ids(self, gathered32, world, Bl, out64, label_out=None):
        out64.copy_(gathered32.view(world, 3, Bl).permute(1, 0, 2).reshape(3, world * Bl).to(torch.int64))

    def presort(self, a, b, c):
        pass                                                            # the stand-in's apply segments the ids itself

    def partial(self, uid, pid, nid, diff):
        u, p, n = self.U[uid], self.I[pid], self.I[nid]
        B = uid.numel()
        diff[:B] = (u * p).sum(1) - (u * n).sum(1)
        diff[B] = (u * u).sum()
        diff[B + 1] = (p * p).sum()

    def grad_apply(self, uid, pid, nid, diff):
        B = uid.numel()
        u, p, n = self.U[uid], self.I[pid], self.I[nid]
        s = torch.sigmoid(diff[:B])
        g = -(1.0 / B) * (s * (1 - s)) / (self.gamma + s)
        GU, GP = g[:, None] * (p - n), g[:, None] * u
        OracleOps().finish_sums(torch.stack([(-torch.log(self.gamma + s)).sum(), diff[B], diff[B + 1]]), B, self.reg_weight, self.out)
        self.t += 1
        o = OracleOps()
        o.sort_apply(self.U, self.state[0], uid, GU, self.opt, self.hp, self.t, reg_limit=B, reg_coef=self.out[4:5])
        o.sort_apply(self.I, self.state[1], torch.cat([pid, nid]), torch.cat([GP, -GP]), self.opt, self.hp, self.t, reg_limit=B,
                     reg_coef=self.out[5:6])
        return self.out


def _worker_dim(rank, world, port, opt, q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        import recbole_cdr_amd  # noqa: F401
        from recbole_cdr_amd.dimshard import DimShardedBPRStep, dim_shard_of, dim_to_row_shards, row_to_dim_shards
        torch.manual_seed(0)
        nu, ni, D, B, reg, lr = 41, 29, 8 * world, 23, 0.03, 0.05
        U, I = torch.randn(nu, D) * 0.3, torch.randn(ni, D) * 0.3
        Uc, Ic = dim_shard_of(U, world, rank), dim_shard_of(I, world, rank)
        hp = {'lr': lr, 'b1': 0.9, 'b2': 0.999, 'eps': 1e-8, 'wd': 0.0}
        st = DimShardedBPRStep(Uc, Ic, B, ops=OracleDimOps(Uc, Ic, 1 if opt == 'adam' else 0, hp, 1e-10, reg))
        losses, batches = [], []
        for step in range(3):
            g = torch.Generator(); g.manual_seed(100 * step + rank)
            u = torch.randint(0, nu, (B,), generator=g); p = torch.randint(0, ni, (B,), generator=g)
            n = torch.randint(0, ni, (B,), generator=g)
            batches.append((u, p, n))
            losses.append(float(st.step(u, p, n)[0]))
        rows = dim_to_row_shards(Uc)                                           # the layout the sharded full-sort evaluates on
        back = row_to_dim_shards(rows, nu)
        assert torch.equal(back, Uc)
        q.put((rank, Uc.numpy().copy(), Ic.numpy().copy(), losses, [tuple(t.numpy().copy() for t in b) for b in batches],
               rows.numpy().copy()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('world,opt', [(2, 'sgd'), (3, 'adam')])
def test_dim_sharded_step_matches_single_process(world, opt):
    """Column slices of the single-process result, the global loss on every rank, and the column -> row-shard transpose."""
    from oracle import train_step as ts
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_dim, args=(r, world, port, opt, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    torch.manual_seed(0)
    nu, ni, D, reg, lr = 41, 29, 8 * world, 0.03, 0.05
    U, I = torch.randn(nu, D) * 0.3, torch.randn(ni, D) * 0.3
    us, is_ = ts.RowwiseAdamState(U), ts.RowwiseAdamState(I)
    for step in range(3):
        u, p, n = (torch.cat([torch.from_numpy(res[r][4][step][k]) for r in range(world)]) for k in range(3))
        loss = ts.rowwise_step(U, I, us, is_, u, p, n, step + 1, opt=opt, lr=lr, reg_weight=reg)
        for r in range(world):
            assert abs(res[r][3][step] - float(loss)) <= 1e-5 * abs(float(loss)), (step, res[r][3][step], float(loss))
    Ds = D // world
    atol = lr * 1e-2 if opt == 'adam' else 1e-6
    for r in range(world):
        torch.testing.assert_close(torch.from_numpy(res[r][1]), U[:, r * Ds:(r + 1) * Ds], rtol=2e-5, atol=atol)
        torch.testing.assert_close(torch.from_numpy(res[r][2]), I[:, r * Ds:(r + 1) * Ds], rtol=2e-5, atol=atol)
        torch.testing.assert_close(torch.from_numpy(res[r][5]), U[r::world], rtol=2e-5, atol=atol)


class OraclePointDimOps(OracleDimOps):
    """Pointwise rows (user, item, label), MSE on the dot: the label travels as its fp32 bit pattern in the third int32 slot."""

    def pack_ids(self, uid, iid, label, out32):
        out32.copy_(torch.cat([uid.to(torch.int32), iid.to(torch.int32), label.view(torch.int32)]))

    def unpack_ids(self, gathered32, world, Bl, out64, label_out=None):
        g = gathered32.view(world, 3, Bl).permute(1, 0, 2).reshape(3, world * Bl)
        out64[:2].copy_(g[:2].to(torch.int64))
        label_out.copy_(g[2].contiguous().view(torch.float32))

    def partial(self, uid, iid, label, dot):
        u, v = self.U[uid], self.I[iid]
        B = uid.numel()
        dot[:B] = (u * v).sum(1)
        dot[B] = (u * u).sum()
        dot[B + 1] = (v * v).sum()

    def grad_apply(self, uid, iid, label, dot):
        B = uid.numel()
        u, v = self.U[uid], self.I[iid]
        d = dot[:B] - label
        g = 2.0 * d / B
        OracleOps().finish_sums(torch.stack([(d * d).sum(), dot[B], dot[B + 1]]), B, self.reg_weight, self.out)
        self.t += 1
        o = OracleOps()
        o.sort_apply(self.U, self.state[0], uid, g[:, None] * v, self.opt, self.hp, self.t, reg_limit=B, reg_coef=self.out[4:5])
        o.sort_apply(self.I, self.state[1], iid, g[:, None] * u, self.opt, self.hp, self.t, reg_limit=B, reg_coef=self.out[5:6])
        return self.out


def _worker_dim_point(rank, world, port, q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        import recbole_cdr_amd  # noqa: F401
        from recbole_cdr_amd.dimshard import DimShardedPointStep, dim_shard_of
        torch.manual_seed(0)
        nu, ni, D, B, reg, lr = 41, 29, 8 * world, 23, 0.03, 0.05
        U, I = torch.randn(nu, D) * 0.3, torch.randn(ni, D) * 0.3
        Uc, Ic = dim_shard_of(U, world, rank), dim_shard_of(I, world, rank)
        hp = {'lr': lr, 'b1': 0.9, 'b2': 0.999, 'eps': 1e-8, 'wd': 0.0}
        st = DimShardedPointStep(Uc, Ic, B, ops=OraclePointDimOps(Uc, Ic, 1, hp, 0.0, reg))
        losses, batches = [], []
        for step in range(3):
            g = torch.Generator(); g.manual_seed(100 * step + rank)
            u = torch.randint(0, nu, (B,), generator=g); i = torch.randint(0, ni, (B,), generator=g)
            y = (torch.rand(B, generator=g) < 0.4).float()
            batches.append((u, i, y))
            losses.append(float(st.step(u, i, y)[0]))
        q.put((rank, Uc.numpy().copy(), Ic.numpy().copy(), losses, [tuple(t.numpy().copy() for t in b) for b in batches]))
    finally:
        dist.destroy_process_group()


def test_dim_sharded_point_step_matches_single_process():
    from oracle import train_step as ts
    world = 2
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_dim_point, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    torch.manual_seed(0)
    nu, ni, D, reg, lr = 41, 29, 8 * world, 0.03, 0.05
    U, I = torch.randn(nu, D) * 0.3, torch.randn(ni, D) * 0.3
    us, is_ = ts.RowwiseAdamState(U), ts.RowwiseAdamState(I)
    for step in range(3):
        u, i, y = (torch.cat([torch.from_numpy(res[r][4][step][k]) for r in range(world)]) for k in range(3))
        loss = ts.rowwise_point_step(U, I, us, is_, u, i, y, step + 1, step + 1, loss='mse', opt='adam', lr=lr, reg_weight=reg)
        for r in range(world):
            assert abs(res[r][3][step] - float(loss)) <= 1e-5 * abs(float(loss)), (step, res[r][3][step], float(loss))
    Ds = D // world
    for r in range(world):
        torch.testing.assert_close(torch.from_numpy(res[r][1]), U[:, r * Ds:(r + 1) * Ds], rtol=2e-5, atol=lr * 1e-2)
        torch.testing.assert_close(torch.from_numpy(res[r][2]), I[:, r * Ds:(r + 1) * Ds], rtol=2e-5, atol=lr * 1e-2)


def _worker_cols(rank, world, port, q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        import recbole_cdr_amd  # noqa: F401
        from recbole_cdr_amd.dimshard import cols_to_row_shards, state_cols_to_row_shards
        from recbole_cdr_amd.fused import RowwiseState
        torch.manual_seed(4)
        rows, D = 37, 24
        full, m, v = torch.randn(rows, D), torch.randn(rows, D), torch.rand(rows, D)
        out = {}
        for name, holders in (('lower', [0, 1]), ('upper', [2, 3]), ('one', [3])):
            Ds = D // len(holders)
            st = None
            if rank in holders:
                j = holders.index(rank)
                st = RowwiseState.__new__(RowwiseState)
                st.step = 5
                st.table, st.exp_avg, st.exp_avg_sq = (t[:, j * Ds:(j + 1) * Ds].contiguous() for t in (full, m, v))
            new = state_cols_to_row_shards(st, rows, Ds, holders, True)
            assert new.step == 5 and (st is None or st.table is None)
            out[name] = tuple(t.numpy().copy() for t in (new.table, new.exp_avg, new.exp_avg_sq))
        only = cols_to_row_shards(full[:, :12].contiguous() if rank == 1 else None, rows, 12, [1])
        # and back: row shards on every rank -> column blocks on a (different) holder set, nothing elsewhere
        from recbole_cdr_amd.dimshard import row_shards_to_cols
        back = row_shards_to_cols(full[rank::world].contiguous(), rows, D, [0, 2, 3])
        assert (back is None) == (rank == 1)
        if back is not None:
            j = [0, 2, 3].index(rank)
            assert torch.equal(back, full[:, j * 8:(j + 1) * 8])
        q.put((rank, out, only.numpy().copy()))
    finally:
        dist.destroy_process_group()


def test_cols_to_row_shards_from_a_subset_of_ranks():
    """A table held (in column blocks) by half of the ranks -- one domain's group -- becomes row shards on ALL ranks, moments and
    update count included: the phase switch of the domain-group layout (bench.py --shard dim)."""
    world = 4
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_cols, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    torch.manual_seed(4)
    rows, D = 37, 24
    full, m, v = torch.randn(rows, D), torch.randn(rows, D), torch.rand(rows, D)
    for r in range(world):
        for name in ('lower', 'upper', 'one'):
            for got, want in zip(res[r][1][name], (full, m, v)):
                assert torch.equal(torch.from_numpy(got), want[r::world]), (r, name)
        assert torch.equal(torch.from_numpy(res[r][2]), full[r::world, :12])


def _worker_ragged(rank, world, port, q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        import recbole_cdr_amd  # noqa: F401
        from recbole_cdr_amd.dimshard import DimShardedBPRStep
        U, I = torch.randn(30, 8), torch.randn(20, 8)
        hp = {'lr': 0.1, 'b1': 0.9, 'b2': 0.999, 'eps': 1e-8, 'wd': 0.0}
        st = DimShardedBPRStep(U, I, 32, ops=OracleDimOps(U, I, 0, hp, 1e-10, 0.0))
        B = 23 + rank                                                        # one row more on rank 1
        try:
            st.step(torch.randint(0, 30, (B,)), torch.randint(0, 20, (B,)), torch.randint(0, 20, (B,)))
            q.put((rank, 'no error'))
        except ValueError as e:
            q.put((rank, str(e)))
    finally:
        dist.destroy_process_group()


def test_dim_sharded_step_rejects_ragged_ranks():
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_ragged, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(2)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
    assert all('same number of rows on every rank, got [23, 24]' in r[1] for r in res), res


# ---------------------------------------------------------------------------------------------- row-sharded BiTGCF (configs[3])
class OracleGraphOps:
    """Stand-in local arithmetic for bitgcf_shard.ShardedBiTGCF: the reference's formulas in torch on CPU tensors (graph layer
    bitgcf.py:130-135, transfer layer :137-172, F.normalize, BCE + EmbLoss :221-247); backward pieces through torch.autograd
    of those same formulas."""

    def tensor(self, a, dtype=None):
        return torch.as_tensor(a)

    @staticmethod
    def _adj(csr, ncols):
        indptr, indices, values = csr
        return torch.sparse_csr_tensor(indptr, indices, values, (indptr.numel() - 1, ncols))

    def graph_layer_fwd(self, csr, Eg, E):
        side = torch.sparse.mm(self._adj(csr, Eg.shape[0]), Eg)
        return side, E + (side + E * side)

    def mul_one_plus(self, g, x):
        return g * (1.0 + x)

    def graph_layer_bwd(self, csr, tmp_g, gnew, side):
        return gnew * (1.0 + side) + torch.sparse.mm(self._adj(csr, tmp_g.shape[0]), tmp_g)

    @staticmethod
    def _transfer(S, T, ds, dt, n_overlap, lam_s, lam_t):
        ds, dt = ds.view(-1, 1), dt.view(-1, 1)
        lap = (ds * S + dt * T) / ((ds + dt) + 1e-7)
        s_lam = lam_s * S + (1 - lam_s) * T
        t_lam = lam_t * T + (1 - lam_t) * S
        So = torch.cat([((s_lam + lap) / 2)[:n_overlap], S[n_overlap:]], 0)
        To = torch.cat([((t_lam + lap) / 2)[:n_overlap], T[n_overlap:]], 0)
        return So, To

    def transfer_fwd(self, S, T, ds, dt, n_overlap, lam_s, lam_t):
        return self._transfer(S, T, ds, dt, n_overlap, lam_s, lam_t)

    def transfer_bwd(self, gSo, gTo, ds, dt, n_overlap, lam_s, lam_t):
        S = torch.zeros_like(gSo, requires_grad=True); T = torch.zeros_like(gTo, requires_grad=True)      # the layer is linear
        So, To = self._transfer(S, T, ds, dt, n_overlap, lam_s, lam_t)
        return torch.autograd.grad([So, To], [S, T], [gSo, gTo])

    def l2norm_fwd(self, x):
        return torch.nn.functional.normalize(x, p=2, dim=1), x.norm(dim=1)

    def l2norm_bwd(self, x, nrm, gy):
        xr = x.clone().requires_grad_(True)
        return torch.autograd.grad(torch.nn.functional.normalize(xr, p=2, dim=1), xr, gy)[0]

    def gather_owned(self, X, pos, lo):
        q = pos - lo
        own = (pos >= 0) & (q >= 0) & (q < X.shape[0])
        out = torch.zeros(pos.numel(), X.shape[1], dtype=X.dtype)
        out[own] = X[q[own]]
        return out

    def scatter_owned(self, rows, W, pos, lo, src):
        q = pos - lo
        own = (pos >= 0) & (q >= 0) & (q < rows)
        return torch.zeros(rows, W, dtype=src.dtype).index_add_(0, q[own], src[own])

    def slice_partials(self, rows, Bs, n, W, D, concat, label):
        if n == 0:
            return None, torch.zeros(3)
        x = rows.detach().clone().requires_grad_(True)
        xu, xi = x[:n, :W], x[Bs:Bs + n, :W]
        eu, ei = (x[:n, :D], x[Bs:Bs + n, :D]) if concat else (x[:n, W:], x[Bs:Bs + n, W:])
        p = torch.sigmoid((xu * xi).sum(1))
        bce_sum = torch.nn.functional.binary_cross_entropy(p, label, reduction='sum')        # recbole's BCELoss, un-averaged
        return (x, eu, ei, bce_sum), torch.stack([bce_sum.detach(), (eu.detach() ** 2).sum(), (ei.detach() ** 2).sum()])

    def slice_grads(self, state, totals, B, reg_weight):
        x, eu, ei, bce_sum = state
        # d/d rows of [BCE-sum / B + reg_weight (||Eu||_F + ||Ei||_F) / B] with the norms of the WHOLE batch (EmbLoss: un-squared norms):
        # d||E||_F / d e = e / ||E||_F, written as a surrogate whose autograd gradient is exactly that
        nu, ni = torch.sqrt(totals[1]), torch.sqrt(totals[2])
        surrogate = bce_sum / B + reg_weight * ((eu * eu.detach()).sum() / nu + (ei * ei.detach()).sum() / ni) / B
        return torch.autograd.grad(surrogate, x)[0]

    def batch_loss(self, out_g, E0_g, pu, pi, label, reg_weight):
        from oracle.losses import bce_loss, emb_loss
        a = out_g.detach().requires_grad_(True); e = E0_g.detach().requires_grad_(True)
        p = torch.sigmoid((a[pu] * a[pi]).sum(1))
        loss = bce_loss(p, label) + reg_weight * emb_loss(e[pu], e[pi])
        ga, ge = torch.autograd.grad(loss.sum(), [a, e])
        return loss.detach().reshape(()), ga, ge


def _bitgcf_case(connect_way, seed=3, B=40):
    import numpy as np
    from oracle.common import IdSpace
    ids = IdSpace(OU=9, TOU=7, SOU=5, OI=1, TOI=11, SOI=8) if connect_way == 'concat' else IdSpace(OU=1, TOU=8, SOU=6, OI=10, TOI=7, SOI=9)
    rs = np.random.RandomState(seed)
    nu, ni = ids.total_num_users, ids.total_num_items
    def pairs(users, items, n):
        return np.stack([rs.choice(users, n), rs.choice(items, n)], 1).astype(np.int64)
    su = np.r_[np.arange(1, ids.OU), np.arange(ids.OU + ids.TOU, nu)]
    si = np.r_[np.arange(1, ids.OI), np.arange(ids.OI + ids.TOI, ni)]
    s_pairs = pairs(su, si, 90)
    t_pairs = pairs(np.arange(1, ids.OU + ids.TOU), np.arange(1, ids.OI + ids.TOI), 110)
    g = torch.Generator().manual_seed(seed)
    D = 8
    params = {k: torch.randn(nu if '_user_' in k else ni, D, generator=g) * 0.3
              for k in ('source_user_embedding.weight', 'source_item_embedding.weight', 'target_user_embedding.weight', 'target_item_embedding.weight')}
    inter = {'source_user_id': torch.from_numpy(s_pairs[:B, 0].copy()), 'source_item_id': torch.from_numpy(s_pairs[:B, 1].copy()),
             'source_label': (torch.rand(B, generator=g) < 0.5).float(),
             'target_user_id': torch.from_numpy(t_pairs[:B, 0].copy()), 'target_item_id': torch.from_numpy(t_pairs[:B, 1].copy()),
             'target_label': (torch.rand(B, generator=g) < 0.5).float()}
    return ids, s_pairs, t_pairs, params, inter, D


def _worker_bitgcf(rank, world, port, connect_way, q, batch_loss='routed', B=40):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        import recbole_cdr_amd  # noqa: F401
        from recbole_cdr_amd.bitgcf_shard import ShardedBiTGCF
        ids, s_pairs, t_pairs, params, inter, D = _bitgcf_case(connect_way, B=B)
        m = ShardedBiTGCF(ids.total_num_users, ids.total_num_items, ids.OU, ids.OI, s_pairs, t_pairs, D, 2, 0.8, 0.7, connect_way, 0.01,
                          OracleGraphOps(), init=params, batch_loss=batch_loss)
        opt = torch.optim.Adam(list(m.params.values()), lr=0.01)
        losses = []
        for _ in range(2):                                        # two steps: the second one runs on updated shards
            opt.zero_grad()
            ls, lt = m.loss_and_grads(inter)
            losses.append((float(ls), float(lt)))
            opt.step()
        full = m.full_tables()
        prop = m.propagated_tables()
        q.put((rank, losses, {k: v.numpy().copy() for k, v in full.items()}, [t.numpy().copy() for t in prop]))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('world,connect_way,batch_loss,B', [(2, 'concat', 'routed', 40), (3, 'mean', 'routed', 40), (4, 'concat', 'routed', 40),
                                                            (3, 'concat', 'routed', 40), (4, 'mean', 'routed', 5), (2, 'mean', 'replicated', 40),
                                                            (4, 'concat', 'replicated', 40)])
def test_row_sharded_bitgcf_matches_single_process(world, connect_way, batch_loss, B):
    """BASELINE configs[3]: tables, Adam state, adjacency rows and transfer-layer degrees cut into ``world`` row blocks, per-layer
    all-gather of E (forward) and of g (1 + E) (backward); uneven blocks (padding rows), overlap rows that straddle the block
    boundary, user-overlap and item-overlap id spaces.  Two Adam steps: both losses, all four full tables and the propagated
    tables equal the single-process oracle (torch autograd over the reference's formulas).  ``routed``: rank r scores its B / world
    slice of the batch (40 rows over 3 ranks: 14 + 14 + 12, padding slots) on rows delivered by a reduce-scatter, the BCE sum and the
    EmbLoss sums of squares are all-reduced, gradient rows return by all-gather; ``replicated``: every rank scores the whole batch on the
    all-gathered stacked tables.  B = 5 over 4 ranks: slices of 2, 2, 1 and 0 rows (a rank with nothing to score still takes part in every collective)."""
    from oracle import bitgcf as obit
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_bitgcf, args=(r, world, port, connect_way, q, batch_loss, B)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=180) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    ids, s_pairs, t_pairs, params, inter, D = _bitgcf_case(connect_way, B=B)
    ref = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    graph = obit.build_graph(s_pairs, t_pairs, ids.total_num_users, ids.total_num_items)
    opt = torch.optim.Adam(list(ref.values()), lr=0.01)
    want = []
    for _ in range(2):
        opt.zero_grad()
        ls, lt = obit.calculate_loss(ref, ids, graph, inter, 2, 0.8, 0.7, connect_way, 0.01)
        (ls + lt).sum().backward()
        want.append((float(ls), float(lt)))
        opt.step()
    with torch.no_grad():
        prop = obit.forward(ref, ids, graph, 2, 0.8, 0.7, connect_way)
    for r, losses, full, got_prop in res:
        for (a, b), (c, d) in zip(losses, want):
            assert abs(a - c) <= 1e-5 * abs(c) and abs(b - d) <= 1e-5 * abs(d), (r, losses, want)
        for k, v in ref.items():
            torch.testing.assert_close(torch.from_numpy(full[k]), v.detach(), rtol=1e-5, atol=0.01 * 1e-2)
        for a, b in zip(got_prop, prop):
            torch.testing.assert_close(torch.from_numpy(a), b, rtol=1e-5, atol=1e-6)


# ---------------------------------------------------------------------------------------------------------------------
# bench.py --gpus N for C5, as the package lays it out (recbole_cdr_amd/c5_layouts.py: the SAME group creation, step objects,
# prefetched id exchange and two-domain pipelining bench.py runs), under gloo on CPU with stand-in arithmetic.
def _worker_layout(rank, world, port, layout, q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        import recbole_cdr_amd  # noqa: F401
        from recbole_cdr_amd import c5_layouts
        from recbole_cdr_amd.dimshard import dim_shard_of
        from recbole_cdr_amd.shard import shard_of
        nu, ni, D, B, reg, lr = 47, 31, 16, 11, 0.03, 0.05
        torch.manual_seed(0)
        full = {k: torch.randn(r, D) * 0.3 for k, r in (('su', nu), ('si', ni), ('tu', nu), ('ti', ni))}
        mode, Ds = c5_layouts.resolve(world, layout, D)

        def make_table(name, rows, cols, total_cols):
            if mode == 'row':
                t = shard_of(full[name], world, rank)
            else:
                G = world // 2 if mode == 'dim-groups' else world
                t = dim_shard_of(full[name], G, rank % G)
            assert tuple(t.shape) == (rows, cols), (name, t.shape, rows, cols)
            return t.clone()
        hp = {'lr': lr, 'b1': 0.9, 'b2': 0.999, 'eps': 1e-8, 'wd': 0.0}
        lay = c5_layouts.build(world, rank, layout, D, B, nu, ni, make_table, dict(opt='adam', lr=lr, reg_weight=reg), device='cpu',
                               dim_ops=lambda U, I, mb: OracleDimOps(U, I, 1, hp, 1e-10, reg), row_ops=OracleOps)
        assert lay.mode == mode
        doms = lay.rank_domains()
        nb = 2 * B if mode == 'dim-groups' else B
        pool = []
        for it in range(3):
            b = {}
            for d in doms:
                g = torch.Generator(); g.manual_seed(1000 * it + 10 * (d == 'target') + rank)
                b[d] = (torch.randint(0, nu, (nb,), generator=g), torch.randint(0, ni, (nb,), generator=g), torch.randint(0, ni, (nb,), generator=g))
            pool.append(b)
        losses = []
        for i in range(4):                                       # step 3 re-uses batch 0: the prefetch slot wraps round the pool
            lay.run(pool, i)
            losses.append({d: float(lay.steps[d].loss_value()) for d in doms})
        q.put((rank, mode, {k: v.numpy().copy() for k, v in lay.tabs.items()}, losses,
               [{d: tuple(t.numpy().copy() for t in b[d]) for d in b} for b in pool]))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('world,layout', [(2, 'dim'), (4, 'dim'), (8, 'dim'), (2, 'row'), (4, 'row'), (8, 'row')])
def test_bench_layouts_under_gloo(world, layout):
    """N = 2: the dimension layout cuts BOTH domains over both ranks (collectives at the first multi-GPU point); N = 4: one domain
    per half of the ranks with the prefetched id all-gather; 'row': the pipelined all-to-all exchange.  Losses of every step and
    every rank's slice of all four tables equal the single-process oracle on the concatenated batches."""
    from oracle import train_step as ts
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_layout, args=(r, world, port, layout, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=240) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    mode = res[0][1]
    assert mode == {(2, 'dim'): 'dim', (4, 'dim'): 'dim-groups', (8, 'dim'): 'dim-groups', (2, 'row'): 'row', (4, 'row'): 'row', (8, 'row'): 'row'}[(world, layout)]
    nu, ni, D, reg, lr = 47, 31, 16, 0.03, 0.05
    torch.manual_seed(0)
    full = {k: torch.randn(r, D) * 0.3 for k, r in (('su', nu), ('si', ni), ('tu', nu), ('ti', ni))}
    half = world // 2
    for d in ('source', 'target'):
        members = [r for r in range(world) if mode != 'dim-groups' or (r < half) == (d == 'source')]
        U, I = full[d[0] + 'u'], full[d[0] + 'i']
        us, is_ = ts.RowwiseAdamState(U), ts.RowwiseAdamState(I)
        for i in range(4):
            u, p, n = (torch.cat([torch.from_numpy(res[r][4][i % 3][d][k]) for r in members]) for k in range(3))
            loss = ts.rowwise_step(U, I, us, is_, u, p, n, i + 1, opt='adam', lr=lr, reg_weight=reg)
            for r in members:
                assert abs(res[r][3][i][d] - float(loss)) <= 1e-5 * abs(float(loss)), (d, i, r, res[r][3][i][d], float(loss))
        for gi, r in enumerate(members):
            for name, T in ((d[0] + 'u', U), (d[0] + 'i', I)):
                got = torch.from_numpy(res[r][2][name])
                if mode == 'row':
                    want = T[r::world]
                else:
                    G = len(members)
                    want = T[:, gi * (D // G):(gi + 1) * (D // G)]
                torch.testing.assert_close(got, want, rtol=2e-5, atol=lr * 1e-2)


# ---- preflight: guarded bring-up of a layout and the fallback chain ------------------------------------------------
def _worker_preflight(rank, world, port, inject, q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    os.environ['CDR_PREFLIGHT_FAIL'] = inject
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        import recbole_cdr_amd  # noqa: F401
        from recbole_cdr_amd import c5_layouts, preflight
        from recbole_cdr_amd.dimshard import dim_shard_of
        from recbole_cdr_amd.shard import shard_of
        ctrl = preflight.control_group(timeout_s=60)
        nu, ni, D, B, reg, lr = 47, 31, 16, 11, 0.03, 0.05
        torch.manual_seed(0)
        full = {k: torch.randn(r, D) * 0.3 for k, r in (('su', nu), ('si', ni), ('tu', nu), ('ti', ni))}
        hp = {'lr': lr, 'b1': 0.9, 'b2': 0.999, 'eps': 1e-8, 'wd': 0.0}

        all_groups = {name: c5_layouts.make_groups(world, name) for name in ('dim-groups', 'dim', 'row')}

        def build(name):
            mode = name

            def make_table(nm, rows, cols, total_cols):
                if mode == 'row':
                    return shard_of(full[nm], world, rank).clone()
                G = world // 2 if mode == 'dim-groups' else world
                return dim_shard_of(full[nm], G, rank % G).clone()
            lay = c5_layouts.build(world, rank, 'row' if name == 'row' else 'dim', D, B, nu, ni, make_table, dict(opt='adam', lr=lr, reg_weight=reg),
                                   domain_groups=name == 'dim-groups', device='cpu', groups=all_groups[name],
                                   dim_ops=lambda U, I, mb: OracleDimOps(U, I, 1, hp, 1e-10, reg), row_ops=OracleOps)
            assert lay.mode == name
            nb = 2 * B if name == 'dim-groups' else B
            lay.batches = []
            for it in range(2):
                b = {}
                for d in lay.rank_domains():
                    g = torch.Generator(); g.manual_seed(1000 * it + 10 * (d == 'target') + rank)
                    b[d] = (torch.randint(0, nu, (nb,), generator=g), torch.randint(0, ni, (nb,), generator=g), torch.randint(0, ni, (nb,), generator=g))
                lay.batches.append(b)
            return lay

        def first_steps(lay):
            for i in range(2):
                lay.run(lay.batches, i)
        name, lay, attempts = preflight.try_layouts(['dim-groups', 'dim', 'row'], build, first_steps, ctrl, seconds=8.0, device='cpu')
        seen = None
        if lay is not None:
            seen = {d: preflight.ranks_seen(g, 'cpu') for d, g in lay.groups.items() if d in lay.rank_domains()}
            lay.run(lay.batches, 0)                                   # the layout that came up keeps working
        q.put((rank, name, [(a['layout'], a['ok'], sorted(a['errors'])) for a in attempts], seen))
    finally:
        if inject:
            q.close(); q.join_thread()                               # (the result has left this process before it is cut short)
            os._exit(0)                                              # (watchdog threads are still blocked in abandoned collectives: as bench.py does)
        dist.destroy_process_group()


@pytest.mark.parametrize('inject,used,tried', [
    ('', 'dim-groups', [('dim-groups', True, [])]),
    # (rank 1 raises before its first collective: its group partner, rank 0, waits for it until the deadline and is abandoned too)
    ('dim-groups:raise@1', 'dim', [('dim-groups', False, [0, 1]), ('dim', True, [])]),
    ('dim-groups:hang@2,dim:raise', 'row', [('dim-groups', False, [2, 3]), ('dim', False, [0, 1, 2, 3]), ('row', True, [])]),
    ('dim-groups,dim,row:raise@3', None, [('dim-groups', False, [0, 1, 2, 3]), ('dim', False, [0, 1, 2, 3]), ('row', False, [0, 1, 2, 3])]),
])
def test_preflight_fallback_chain_under_gloo(inject, used, tried):
    """bench.py --gpus N brings its layout up under recbole_cdr_amd/preflight.py: a candidate that raises on one rank, or blocks past
    the deadline on one rank, is abandoned by EVERY rank (verdicts agreed over the gloo control group) and the next one is tried --
    dim-groups -> dim -> row -> none (the caller then runs independent replicas).  World 4 on CPU with the oracle arithmetic."""
    world = 4
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_preflight, args=(r, world, port, inject, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=240) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for r, name, attempts, seen in res:
        assert name == used, (r, name, attempts)
        assert attempts == tried, (r, attempts)
        if used == 'dim-groups':
            assert list(seen.values()) == [2]
        elif used is not None:
            assert seen == {'source': 4, 'target': 4}
