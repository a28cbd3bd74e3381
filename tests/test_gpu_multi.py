"""GPU parity (-m gpu), multi-GPU forms: the dimension-sharded BPR / MF step (csrc/cdr_dimshard.hip, dimshard.py), the layout
transposes, the sharded evaluation, CrossDomainTrainer with config['dist_group'] / parallel_domains, sharded checkpoints, and
bench.py launched over several ranks.  Several ranks share cuda:0 and talk over gloo (RCCL refuses two ranks on one device);
the one-rank RCCL test issues the same collectives on a real communicator."""
import numpy as np
import pytest
import torch

from helpers import DEV, FakeDataset, base_config, assert_close, load_params, make_mapping as _make_mapping, collect_ranks as _collect_ranks

pytestmark = pytest.mark.gpu


# ---------------------------------------------------------------------------------------------------------------------
# dimension-sharded BPR step (dimshard.py, csrc/cdr_dimshard.hip)
@pytest.mark.parametrize('D,B', [(128, 5000), (64, 333), (32, 1), (16, 4097), (8, 77)])
def test_partial_diff_and_grad_from_diff_equal_the_fused_forward(D, B):
    """One rank holding ALL columns: cdr_bpr_partial_diff + cdr_bpr_grad_from_diff reproduce cdr_bpr_fwd_grad (loss,
    norms, coefficients, compact gradient rows) -- the arithmetic that the all-reduce is cut into."""
    from recbole_cdr_amd import binding as B_
    torch.manual_seed(D + B)
    nu, ni = 900, 700
    U, I = torch.randn(nu, D, device=DEV) * 0.3, torch.randn(ni, D, device=DEV) * 0.3
    u, p, n = torch.randint(0, nu, (B,), device=DEV), torch.randint(0, ni, (B,), device=DEV), torch.randint(0, ni, (B,), device=DEV)
    out_a, GU_a, GP_a = torch.zeros(12, device=DEV), torch.empty(B, D, device=DEV), torch.empty(B, D, device=DEV)
    out_b, GU_b, GP_b = torch.zeros(12, device=DEV), torch.empty(B, D, device=DEV), torch.empty(B, D, device=DEV)
    ctx, s = B_.ctx(U.device), B_.stream()
    B_.call('cdr_bpr_fwd_grad', ctx, s, B_.f32(U), B_.f32(I), D, B_.i64(u), B_.i64(p), B_.i64(n), B, 0, 1e-10, 0.05, B_.f32(out_a),
            B_.f32(GU_a), B_.f32(GP_a), 0)
    diff = torch.empty(B + 2, device=DEV)
    B_.call('cdr_bpr_partial_diff', ctx, s, B_.f32(U), B_.f32(I), D, B_.i64(u), B_.i64(p), B_.i64(n), B, B_.f32(diff))
    want = (U[u] * I[p]).sum(1) - (U[u] * I[n]).sum(1)
    assert_close(diff[:B], want, rtol=1e-5, atol=1e-6, what='diff')
    B_.call('cdr_bpr_grad_from_diff', ctx, s, B_.f32(U), B_.f32(I), D, B_.i64(u), B_.i64(p), B_.i64(n), B, 1e-10, 0.05, B_.f32(diff),
            B_.f32(out_b), B_.f32(GU_b), B_.f32(GP_b))
    assert_close(GU_b, GU_a, rtol=1e-5, atol=1e-9, what='GU')                  # dot products may contract to FMAs differently
    assert_close(GP_b, GP_a, rtol=1e-5, atol=1e-9, what='GP')
    assert_close(out_b[:9], out_a[:9], rtol=1e-6, atol=0, what='out9')


def _dim_shared_gpu_worker(rank, world, port, q):
    import os
    import torch.distributed as dist
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        import recbole_cdr_amd  # noqa: F401
        from recbole_cdr_amd.dimshard import DimShardedBPRStep, dim_shard_of, dim_to_row_shards
        torch.cuda.set_device(0)
        torch.manual_seed(11)
        nu, ni, D, B = 7001, 3003, 64, 3000
        tabs = [(torch.randn(nu, D) * 0.1, torch.randn(ni, D) * 0.1) for _ in range(2)]
        steps, cols = [], []
        for d, (U, I) in enumerate(tabs):
            Uc, Ic = dim_shard_of(U, world, rank).to(DEV), dim_shard_of(I, world, rank).to(DEV)
            steps.append(DimShardedBPRStep(Uc, Ic, B, opt='adam', lr=0.01, reg_weight=0.02, group=dist.new_group(backend='gloo'),
                                           stream=torch.cuda.Stream()))
            cols.append((Uc, Ic))
        losses, batches = [], []
        for it in range(3):
            per_dom = []
            for d in range(2):
                g = torch.Generator(); g.manual_seed(1000 * it + 10 * d + rank)
                u = torch.randint(0, nu, (B,), generator=g); p = torch.randint(0, ni, (B,), generator=g)
                n = torch.randint(0, ni, (B,), generator=g)
                if it == 1:
                    u[: B // 2] = u[0]; p[: B // 3] = p[0]; n[100:700] = p[0]        # long segments in both tables
                per_dom.append((u, p, n))
            batches.append(per_dom)
        on_dev = [[tuple(t.to(DEV) for t in dom) for dom in it] for it in batches]
        for it in range(3):
            torch.cuda.synchronize()
            for d in range(2):                                                  # no host sync inside: the two domains just queue up
                # domain 0 announces its next batch: that batch's id exchange runs on a side stream under this step's kernels
                nxt = on_dev[it + 1][d] if (d == 0 and it + 1 < 3) else None
                steps[d].step(*on_dev[it][d], next_batch=nxt)
            torch.cuda.synchronize()
            losses.append([float(s.out[0]) for s in steps])
        rows = dim_to_row_shards(cols[0][1])
        q.put((rank, [(a.cpu().numpy(), b.cpu().numpy()) for a, b in cols], losses,
               [[tuple(t.numpy() for t in dom) for dom in it] for it in batches], rows.cpu().numpy()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('world', [2, 4])
def test_dim_sharded_native_ranks_share_one_gpu(world):
    """DimShardedBPRStep with the native kernels, `world` ranks on cuda:0 over gloo, two domains on their own streams:
    every rank's columns equal the single-GPU fused step's on the concatenated batch; the loss is the global one."""
    import socket
    import torch.multiprocessing as mp
    from recbole_cdr_amd.fused import FusedBPRStep
    s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_dim_shared_gpu_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = _collect_ranks(q, procs)
    torch.manual_seed(11)
    nu, ni, D, B = 7001, 3003, 64, 3000
    Ds = D // world
    tabs = [(torch.randn(nu, D) * 0.1, torch.randn(ni, D) * 0.1) for _ in range(2)]
    for d in range(2):
        U, I = tabs[d][0].to(DEV), tabs[d][1].to(DEV)
        ref = FusedBPRStep(U, I, world * B, opt='adam', lr=0.01, reg_weight=0.02)
        for it in range(3):
            u, p, n = (torch.from_numpy(np.concatenate([res[r][3][it][d][k] for r in range(world)])).to(DEV) for k in range(3))
            loss = float(ref.step(u, p, n)[0])
            for r in range(world):
                assert abs(res[r][2][it][d] - loss) <= 2e-6 * abs(loss), (d, it, r, res[r][2][it][d], loss)
        for r in range(world):
            assert_close(torch.from_numpy(res[r][1][d][0]).to(DEV), U[:, r * Ds:(r + 1) * Ds], rtol=2e-5, atol=1e-4, what=f'U dom{d} rank{r}')
            assert_close(torch.from_numpy(res[r][1][d][1]).to(DEV), I[:, r * Ds:(r + 1) * Ds], rtol=2e-5, atol=1e-4, what=f'I dom{d} rank{r}')
            if d == 0:
                assert_close(torch.from_numpy(res[r][4]).to(DEV), I[r::world], rtol=2e-5, atol=1e-4, what=f'row shard rank{r}')


def test_ids_pack32_unpack32_round_trip_and_overflow_flag():
    from recbole_cdr_amd import binding as B_
    G, Bl = 3, 1001
    g = torch.Generator().manual_seed(5)
    per_rank = [[torch.randint(0, 2 ** 31 - 1, (Bl,), generator=g) for _ in range(3)] for _ in range(G)]
    bad = torch.zeros(1, device=DEV, dtype=torch.int32)
    gathered = torch.empty(G, 3 * Bl, device=DEV, dtype=torch.int32)
    for r in range(G):
        u, p, n = (t.to(DEV) for t in per_rank[r])
        B_.call('cdr_ids_pack32', B_.stream(), B_.i64(u), B_.i64(p), B_.i64(n), None, Bl, B_.raw(gathered[r]), B_.raw(bad))
    out = torch.empty(3, G * Bl, device=DEV, dtype=torch.int64)
    B_.call('cdr_ids_unpack32', B_.stream(), B_.raw(gathered), G, Bl, B_.i64(out), None)
    for j in range(3):
        assert torch.equal(out[j].cpu(), torch.cat([per_rank[r][j] for r in range(G)]))
    assert int(bad.item()) == 0
    big = torch.full((Bl,), 2 ** 31, device=DEV, dtype=torch.int64)
    B_.call('cdr_ids_pack32', B_.stream(), B_.i64(big), B_.i64(big), B_.i64(big), None, Bl, B_.raw(gathered[0]), B_.raw(bad))
    assert int(bad.item()) == 1
    # pointwise rows: the label's bit pattern rides in the third slot and comes back as fp32
    lab = [torch.rand(Bl, generator=g) for _ in range(G)]
    for r in range(G):
        B_.call('cdr_ids_pack32', B_.stream(), B_.i64(per_rank[r][0].to(DEV)), B_.i64(per_rank[r][1].to(DEV)), None, B_.f32(lab[r].to(DEV)),
                Bl, B_.raw(gathered[r]), B_.raw(bad))
    out.fill_(-7)
    lab_out = torch.empty(G * Bl, device=DEV)
    B_.call('cdr_ids_unpack32', B_.stream(), B_.raw(gathered), G, Bl, B_.i64(out), B_.f32(lab_out))
    assert torch.equal(lab_out.cpu(), torch.cat(lab)) and torch.equal(out[1].cpu(), torch.cat([per_rank[r][1] for r in range(G)]))
    assert bool((out[2] == -7).all())


@pytest.mark.parametrize('kind', ['mse', 'bce'])
@pytest.mark.parametrize('D,B', [(128, 3000), (16, 1025), (8, 3)])
def test_point_partial_dot_and_grad_from_dot_equal_the_fused_forward(kind, D, B):
    from recbole_cdr_amd import binding as B_
    torch.manual_seed(D + B)
    nu, ni = 900, 700
    code = B_.CDR_LOSS_MSE if kind == 'mse' else B_.CDR_LOSS_BCE
    U, I = torch.randn(nu, D, device=DEV) * 0.3, torch.randn(ni, D, device=DEV) * 0.3
    u, i = torch.randint(0, nu, (B,), device=DEV), torch.randint(0, ni, (B,), device=DEV)
    y = (torch.rand(B, device=DEV) < 0.4).float()
    out_a, GU_a, GI_a = torch.zeros(12, device=DEV), torch.empty(B, D, device=DEV), torch.empty(B, D, device=DEV)
    out_b, GU_b, GI_b = torch.zeros(12, device=DEV), torch.empty(B, D, device=DEV), torch.empty(B, D, device=DEV)
    ctx, s = B_.ctx(U.device), B_.stream()
    B_.call('cdr_point_fwd_grad', ctx, s, code, B_.f32(U), B_.f32(I), D, B_.i64(u), B_.i64(i), B_.f32(y), B, 0.05, B_.f32(out_a),
            B_.f32(GU_a), B_.f32(GI_a))
    dot = torch.empty(B + 2, device=DEV)
    B_.call('cdr_point_partial_dot', ctx, s, B_.f32(U), B_.f32(I), D, B_.i64(u), B_.i64(i), B, B_.f32(dot))
    assert_close(dot[:B], (U[u] * I[i]).sum(1), rtol=1e-5, atol=1e-6, what='dot')
    B_.call('cdr_point_grad_from_dot', ctx, s, code, B_.f32(U), B_.f32(I), D, B_.i64(u), B_.i64(i), B_.f32(y), B, 0.05, B_.f32(dot),
            B_.f32(out_b), B_.f32(GU_b), B_.f32(GI_b))
    assert_close(GU_b, GU_a, rtol=1e-5, atol=1e-9, what='GU')
    assert_close(GI_b, GI_a, rtol=1e-5, atol=1e-9, what='GI')
    assert_close(out_b[:9], out_a[:9], rtol=1e-6, atol=0, what='out9')


def _dim_point_shared_gpu_worker(rank, world, port, q):
    import os
    import torch.distributed as dist
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        import recbole_cdr_amd  # noqa: F401
        from recbole_cdr_amd.dimshard import DimShardedPointStep, dim_shard_of
        torch.cuda.set_device(0)
        torch.manual_seed(13)
        nu, ni, D, B = 5003, 2001, 96, 2500
        U, I = torch.randn(nu, D) * 0.1, torch.randn(ni, D) * 0.1
        Uc, Ic = dim_shard_of(U, world, rank).to(DEV), dim_shard_of(I, world, rank).to(DEV)
        st = DimShardedPointStep(Uc, Ic, B, loss='mse', opt='adam', lr=0.01, reg_weight=0.02)
        losses, batches = [], []
        for it in range(3):
            g = torch.Generator(); g.manual_seed(1000 * it + rank)
            u = torch.randint(0, nu, (B,), generator=g); i = torch.randint(0, ni, (B,), generator=g)
            y = (torch.rand(B, generator=g) < 0.3).float()
            if it == 1:
                u[: B // 2] = u[0]; i[: B // 3] = i[0]
            batches.append((u, i, y))
            st.step(u.to(DEV), i.to(DEV), y.to(DEV))
            losses.append(float(st.out[0]))
        assert st.ops.ids_fit()
        q.put((rank, Uc.cpu().numpy(), Ic.cpu().numpy(), losses, [tuple(t.numpy() for t in b) for b in batches]))
    finally:
        dist.destroy_process_group()


def test_dim_sharded_point_step_ranks_share_one_gpu():
    """DimShardedPointStep (MF latent factor model: MSE on the dot), 3 ranks x 32 columns on cuda:0 over gloo, against
    FusedPointStep on the full tables and the concatenated rows."""
    import socket
    import torch.multiprocessing as mp
    from recbole_cdr_amd.fused import FusedPointStep
    world = 3
    s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_dim_point_shared_gpu_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = _collect_ranks(q, procs)
    torch.manual_seed(13)
    nu, ni, D, B = 5003, 2001, 96, 2500
    Ds = D // world
    U, I = (torch.randn(nu, D) * 0.1).to(DEV), (torch.randn(ni, D) * 0.1).to(DEV)
    ref = FusedPointStep(U, I, world * B, loss='mse', opt='adam', lr=0.01, reg_weight=0.02)
    for it in range(3):
        u, i, y = (torch.from_numpy(np.concatenate([res[r][4][it][k] for r in range(world)])).to(DEV) for k in range(3))
        loss = float(ref.step(u, i, y)[0])
        for r in range(world):
            assert abs(res[r][3][it] - loss) <= 2e-6 * abs(loss), (it, r, res[r][3][it], loss)
    for r in range(world):
        assert_close(torch.from_numpy(res[r][1]).to(DEV), U[:, r * Ds:(r + 1) * Ds], rtol=2e-5, atol=1e-4, what=f'U rank{r}')
        assert_close(torch.from_numpy(res[r][2]).to(DEV), I[:, r * Ds:(r + 1) * Ds], rtol=2e-5, atol=1e-4, what=f'I rank{r}')


def test_dim_sharded_step_on_rccl_world1_equals_fused():
    """The dimension-sharded step's collectives (int32 all-gather into a buffer slice, fp32 all-reduce of a slice, the column ->
    row all-to-all) issued on a real RCCL communicator -- one rank, so they are identities and the result must equal the plain
    fused step bit for bit; what this pins is that RCCL accepts the buffers, dtypes and views the step hands it."""
    import torch.distributed as dist
    from recbole_cdr_amd.dimshard import DimShardedBPRStep, DimShardedPointStep, dim_to_row_shards, row_to_dim_shards
    from recbole_cdr_amd.fused import FusedBPRStep, FusedPointStep
    import socket
    s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
    dist.init_process_group('nccl', init_method=f'tcp://127.0.0.1:{port}', rank=0, world_size=1, device_id=torch.device('cuda', 0))
    try:
        torch.manual_seed(3)
        nu, ni, D, B = 4001, 3001, 64, 5000
        U, I = torch.randn(nu, D, device=DEV) * 0.1, torch.randn(ni, D, device=DEV) * 0.1
        Ua, Ia, Ub, Ib = U.clone(), I.clone(), U.clone(), I.clone()
        a = DimShardedBPRStep(Ua, Ia, B, opt='adam', lr=0.01, reg_weight=0.02, stream=torch.cuda.Stream())
        a.force_collectives = True
        b = FusedBPRStep(Ub, Ib, B, opt='adam', lr=0.01, reg_weight=0.02)
        pa = DimShardedPointStep(Ua, Ia, B, opt='adam', lr=0.01, reg_weight=0.02, user_state=a.ustate, item_state=a.istate)
        pa.force_collectives = True
        pb = FusedPointStep(Ub, Ib, B, opt='adam', lr=0.01, reg_weight=0.02, user_state=b.ustate, item_state=b.istate)
        for it in range(3):
            u, p, n = torch.randint(0, nu, (B,), device=DEV), torch.randint(0, ni, (B,), device=DEV), torch.randint(0, ni, (B,), device=DEV)
            y = (torch.rand(B, device=DEV) < 0.5).float()
            la = a.step(u, p, n); torch.cuda.synchronize()
            lb = b.step(u, p, n)
            assert_close(la[:6], lb[:6], rtol=1e-6, atol=0, what='bpr out')
            la = pa.step(u, p, y); torch.cuda.synchronize()
            lb = pb.step(u, p, y)
            assert_close(la[:6], lb[:6], rtol=1e-6, atol=0, what='point out')
        assert a.ops.ids_fit() and pa.ops.ids_fit()
        assert_close(Ua, Ub, rtol=2e-5, atol=1e-4, what='U')              # Adam, lr 0.01: FMA contraction differs between the kernels
        assert_close(Ia, Ib, rtol=2e-5, atol=1e-4, what='I')
        assert torch.equal(row_to_dim_shards(dim_to_row_shards(Ua, force=True), nu, force=True), Ua)
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize('loss,opt', [('mse', 'adam'), ('bce', 'adam'), ('mse', 'sgd')])
def test_dim_sharded_point_second_half_one_pass_equals_two_pass(loss, opt):
    """DimShardedPointStep at world 1: the forward-and-update pass fed with the given dots (cdr_point_step_presort +
    cdr_point_step_from_dot) against the two-pass form (cdr_point_grad_from_dot -> cdr_rowwise_apply x 2), batches with long
    duplicate segments on both tables; 1e-6 relative on the loss scalars, tables to the Adam contraction tolerance."""
    from recbole_cdr_amd.dimshard import DimShardedPointStep, NativePointDimOps
    torch.manual_seed(5)
    nu, ni, D, B = 3001, 1501, 48, 6000
    U, I = torch.randn(nu, D, device=DEV) * 0.1, torch.randn(ni, D, device=DEV) * 0.1
    tabs = [(U.clone(), I.clone()) for _ in range(2)]
    hp = dict(loss=loss, opt=opt, lr=0.01, reg_weight=0.02)
    steps = [DimShardedPointStep(Ut, It, B, ops=NativePointDimOps(Ut, It, B, fuse_singles=f, **hp)) for (Ut, It), f in zip(tabs, (True, False))]
    assert steps[0].ops.fused and not steps[1].ops.fused
    for it in range(4):
        u, i = torch.randint(0, nu, (B,), device=DEV), torch.randint(0, ni, (B,), device=DEV)
        y = (torch.rand(B, device=DEV) < 0.4).float()
        if it == 1:
            u[:1000] = 7; i[:70] = 3                       # segments beyond the piece threshold and medium ones
        if it == 2:
            u = torch.arange(B, device=DEV) % nu; i = torch.arange(B, device=DEV) % ni
        outs = [st.step(u, i, y).clone() for st in steps]
        assert_close(outs[0][:4], outs[1][:4], rtol=1e-6, atol=0, what=f'out it{it}')
    assert_close(tabs[0][0], tabs[1][0], rtol=2e-5, atol=1e-5, what='U')
    assert_close(tabs[0][1], tabs[1][1], rtol=2e-5, atol=1e-5, what='I')


# ---------------------------------------------------------------------------------------------------------------------
# The whole EMCDR schedule over several ranks: SOURCE and TARGET BPR steps in the dimension layout, the phase switch
# (tables AND Adam moments transposed to row shards, update counts kept), OVERLAP steps in the row layout, sharded top-k.
def _schedule_worker(rank, world, port, q):
    import os
    import torch.distributed as dist
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        import recbole_cdr_amd  # noqa: F401
        from recbole_cdr_amd.dimshard import DimShardedBPRStep, dim_shard_of, dim_to_row_shards, state_to_row_shards
        from recbole_cdr_amd.fused import FusedMapStep
        from recbole_cdr_amd.shard import ShardedFullSort
        torch.cuda.set_device(0)
        nu, ni, D, B, OB = 1201, 901, 64, 700, 90
        torch.manual_seed(31)
        SU, SI, TU, TI = (torch.randn(n, D) * 0.2 for n in (nu, ni, nu, ni))
        cols = {k: dim_shard_of(t, world, rank).to(DEV) for k, t in (('su', SU), ('si', SI), ('tu', TU), ('ti', TI))}
        hp = dict(opt='adam', lr=0.01, reg_weight=0.02)
        steps = {'source': DimShardedBPRStep(cols['su'], cols['si'], B, **hp), 'target': DimShardedBPRStep(cols['tu'], cols['ti'], B, **hp)}
        losses = []
        for dom, n_steps in (('source', 3), ('target', 2)):                  # unequal counts: the two user tables' Adam step counts differ
            for it in range(n_steps):
                g = torch.Generator(); g.manual_seed({'source': 100, 'target': 200}[dom] + 10 * it + rank)
                u, p, n = (torch.randint(1, hi, (B,), generator=g).to(DEV) for hi in (nu, ni, ni))
                losses.append(float(steps[dom].step(u, p, n)[0]))
        sst, tst = state_to_row_shards(steps['source'].ustate), state_to_row_shards(steps['target'].ustate)
        ti_rows = dim_to_row_shards(cols['ti'])
        _cpu, dev, fn = _make_mapping((D, D), 9)
        fm = FusedMapStep(sst.table, tst.table, fn, dev, OB, lr=0.01, group=dist.group.WORLD, source_state=sst, target_state=tst)
        for it in range(3):
            g = torch.Generator(); g.manual_seed(300 + 10 * it + rank)
            idx = torch.randperm(nu - 1, generator=g)[:OB].add(1).reshape(-1, 1)
            losses.append(float(fm.step(idx.to(DEV))))
        fs = ShardedFullSort(ti_rows, ni)
        ids = torch.arange(1, 41)
        tv, tix = fs.topk(fs.user_rows(tst.table, ids.to(DEV)), 10)
        # the SOURCE phase's slab: two row ranges of the source item table, history in concatenated columns
        si_rows = dim_to_row_shards(cols['si'])
        fs2 = ShardedFullSort(si_rows, ni)
        g = torch.Generator(); g.manual_seed(77)
        hc = torch.sort(torch.randint(1, 7 + (ni - 300), (40, 9), generator=g), dim=1).values
        hp = torch.arange(41) * 9
        rv, rix = fs2.topk_ranges(fs2.user_rows(sst.table, ids.to(DEV)), 10, [(0, 7), (300, ni)], hist_indptr=hp.to(DEV),
                                  hist_cols=hc.reshape(-1).contiguous().to(DEV))
        q.put((rank, losses, tst.table.cpu().numpy(), sst.table.cpu().numpy(), tv.cpu().numpy(), tix.cpu().numpy(),
               dev[0].detach().cpu().numpy(), rv.cpu().numpy(), rix.cpu().numpy(), hc.numpy()))
    finally:
        dist.destroy_process_group()


def test_whole_schedule_dim_then_row_layout_matches_single_gpu():
    import socket
    import torch.multiprocessing as mp
    from recbole_cdr_amd import functional as F_
    from recbole_cdr_amd.fused import FusedBPRStep, FusedMapStep
    world = 2
    s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_schedule_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = _collect_ranks(q, procs)
    nu, ni, D, B, OB = 1201, 901, 64, 700, 90
    torch.manual_seed(31)
    SU, SI, TU, TI = ((torch.randn(n, D) * 0.2).to(DEV) for n in (nu, ni, nu, ni))
    hp = dict(opt='adam', lr=0.01, reg_weight=0.02)
    steps = {'source': FusedBPRStep(SU, SI, world * B, **hp), 'target': FusedBPRStep(TU, TI, world * B, **hp)}
    want = []
    for dom, n_steps in (('source', 3), ('target', 2)):
        for it in range(n_steps):
            parts = []
            for r in range(world):
                g = torch.Generator(); g.manual_seed({'source': 100, 'target': 200}[dom] + 10 * it + r)
                parts.append([torch.randint(1, hi, (B,), generator=g) for hi in (nu, ni, ni)])
            u, p, n = (torch.cat([parts[r][k] for r in range(world)]).to(DEV) for k in range(3))
            want.append(float(steps[dom].step(u, p, n)[0]))
    _cpu, dev, fn = _make_mapping((D, D), 9)
    fm = FusedMapStep(SU, TU, fn, dev, world * OB, lr=0.01, source_state=steps['source'].ustate, target_state=steps['target'].ustate)
    for it in range(3):
        idx = []
        for r in range(world):
            g = torch.Generator(); g.manual_seed(300 + 10 * it + r)
            idx.append(torch.randperm(nu - 1, generator=g)[:OB].add(1))
        want.append(float(fm.step(torch.cat(idx).reshape(-1, 1).to(DEV))))
    for r in range(world):
        for k, (a, b) in enumerate(zip(res[r][1], want)):
            assert abs(a - b) <= 1e-5 * abs(b), (r, k, a, b)
        assert_close(torch.from_numpy(res[r][2]).to(DEV), TU[r::world], rtol=2e-5, atol=1e-4, what=f'target users rank{r}')
        assert_close(torch.from_numpy(res[r][3]).to(DEV), SU[r::world], rtol=2e-5, atol=1e-4, what=f'source users rank{r}')
        assert_close(torch.from_numpy(res[r][6]).to(DEV), dev[0].detach(), rtol=2e-5, atol=1e-4, what='mapping weight')
    tv, tix = F_.fullsort_topk(TU[1:41].contiguous(), TI, None, k=10)
    for r in range(world):
        assert_close(torch.from_numpy(res[r][4]).to(DEV), tv, rtol=1e-4, atol=1e-5, what='top-k values')
        same = (torch.from_numpy(res[r][5]).to(DEV) == tix).float().mean()
        assert float(same) > 0.97, float(same)                      # tables agree to ~1e-5: a near-tie may swap two neighbours
    # two-range slab (SOURCE phase): against the masked single-GPU matrix over cat(SI[:7], SI[300:])
    full = F_.fullsort_scores(SU[1:41].contiguous(), SI[:7], SI[300:])
    full[:, 0] = -float('inf')
    full.scatter_(1, torch.from_numpy(res[0][9]).to(DEV), -float('inf'))
    want = torch.topk(full, 10, dim=1)
    for r in range(world):
        assert_close(torch.from_numpy(res[r][7]).to(DEV), want.values, rtol=1e-4, atol=1e-5, what='two-range top-k values')
        got_scores = torch.gather(full, 1, torch.from_numpy(res[r][8]).to(DEV))
        assert_close(got_scores, want.values, rtol=1e-4, atol=1e-5, what='two-range top-k columns point at the top scores')


# ---------------------------------------------------------------------------------------------------------------------
# CrossDomainTrainer over several ranks: config['dist_group'] + optimizer_mode='rowwise'
def _dist_trainer_setup(dev, lfm, dist_group=None, parallel=False, modes=None, epochs=None):
    from oracle.common import IdSpace
    from recbole_cdr_amd.model.cross_domain_recommender.emcdr import EMCDR
    from recbole_cdr_amd.trainer import CrossDomainTrainer
    from recbole_cdr_amd.data import CrossDomainDataloader, OverlapDataloader, DomainTrainLoader, FullSortEvalLoader
    from recbole_cdr_amd.utils import InputType
    torch.manual_seed(12)
    ids = IdSpace(OU=21, TOU=15, SOU=18, OI=1, TOI=30, SOI=34)
    D, lr, reg = 16, 0.01, 0.01
    extra = {'dist_group': dist_group, 'parallel_domains': parallel} if dist_group is not None else {}
    cfg = base_config(dev, latent_factor_model=lfm, source_embedding_size=D, target_embedding_size=D, reg_weight=reg,
                      mapping_function='non_linear', mlp_hidden_size=[24], learning_rate=lr, optimizer_mode='rowwise',
                      train_modes=modes or ['SOURCE', 'TARGET', 'OVERLAP', 'TARGET'], epoch_num=epochs or ['2', '1', '2', '1'], source_split=False,
                      eval_step=1, epochs=2, topk=[5], valid_metric='recall@5', **extra)
    model = EMCDR(cfg, FakeDataset(ids)).to(dev)
    rng = np.random.RandomState(0)
    src_u = np.array(list(range(1, ids.OU)) + list(range(ids.OU + ids.TOU, ids.total_num_users)))
    src_i = np.arange(ids.OI + ids.TOI, ids.total_num_items)
    tgt_u, tgt_i = np.arange(1, ids.OU + ids.TOU), np.arange(1, ids.OI + ids.TOI)
    s_inter = {'source_user_id': torch.from_numpy(rng.choice(src_u, 96)), 'source_item_id': torch.from_numpy(rng.choice(src_i, 96))}
    t_inter = {'target_user_id': torch.from_numpy(rng.choice(tgt_u, 80)), 'target_item_id': torch.from_numpy(rng.choice(tgt_i, 80))}
    neg_rng = {'s': np.random.RandomState(1), 't': np.random.RandomState(2)}
    s_sampler = lambda u, i, k: torch.from_numpy(neg_rng['s'].choice(src_i, u.numel() * k)).to(u.device)
    t_sampler = lambda u, i, k: torch.from_numpy(neg_rng['t'].choice(tgt_i, u.numel() * k)).to(u.device)
    it = InputType.PAIRWISE if lfm == 'BPR' else InputType.POINTWISE
    train = CrossDomainDataloader(
        DomainTrainLoader(s_inter, 'source_user_id', 'source_item_id', 'source_label', 'neg_', 32, 1, it, s_sampler),
        DomainTrainLoader(t_inter, 'target_user_id', 'target_item_id', 'target_label', 'neg_', 32, 1, it, t_sampler),
        OverlapDataloader(ids.OU, 10))                              # 21 ids -> batches of 10, 10, 1 (the ragged tail is skipped at world 2)
    ev = rng.choice(tgt_u, 40), rng.choice(tgt_i, 40)
    valid = FullSortEvalLoader('target_user_id', np.stack(ev, 1), np.stack([t_inter['target_user_id'].numpy(), t_inter['target_item_id'].numpy()], 1),
                               ids.OI + ids.TOI, 4 * (ids.OI + ids.TOI), dev)
    # SOURCE-phase evaluation (source_split runs): source item ids are not contiguous -- the loader revokes them to the
    # concatenated columns of cat(W_s[:OI], W_s[TI:]) (dataloader.py:240-247)
    sev = rng.choice(src_u, 30), rng.choice(src_i, 30)
    n_src_cols = ids.OI + ids.SOI
    valid_src = FullSortEvalLoader('source_user_id', np.stack(sev, 1), np.stack([s_inter['source_user_id'].numpy(), s_inter['source_item_id'].numpy()], 1),
                                   n_src_cols, 4 * n_src_cols, dev, revoke=(ids.OI, ids.TOI))
    trainer = CrossDomainTrainer(cfg, model)
    trainer.valid_src = valid_src
    return trainer, model, train, valid


def _dist_trainer_worker(rank, world, port, lfm, q, parallel=False):
    import os
    import torch.distributed as dist
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    import faulthandler
    faulthandler.dump_traceback_later(570, exit=True)                  # a wedged collective shows where, instead of hanging the suite
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        import recbole_cdr_amd  # noqa: F401
        torch.cuda.set_device(0)
        trainer, model, train, valid = _dist_trainer_setup(DEV, lfm, dist_group=True, parallel=parallel)
        log = []
        orig = trainer._train_epoch
        trainer._train_epoch = lambda data, e: (log.append(orig(data, e)) or log[-1])
        trainer.fit(train, None, verbose=False, saved=False)
        final = trainer.evaluate(valid)                                      # fit leaves the model in the OVERLAP phase: mapped users
        model.set_phase('TARGET')
        score = trainer.evaluate(valid)['recall@5']
        model.set_phase('SOURCE')
        final = dict(final, source_recall=trainer.evaluate(trainer.valid_src)['recall@5'])
        full = {k: v.cpu().numpy() for k, v in model.gather_full_tables().items()}
        with pytest.raises(RuntimeError, match='whole tables'):            # a shard is not a table: no silent use of one
            model.full_sort_predict(next(iter(valid))[0])
        faulthandler.cancel_dump_traceback_later()
        q.put((rank, log, score, final, full, {k: v.detach().cpu().numpy() for k, v in model.mapping.named_parameters()},
               [tuple(model.source_user_embedding.weight.shape), model._dist.layout('target_user_embedding')]))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('lfm,world,parallel', [('BPR', 2, False), ('MF', 2, False), ('BPR', 4, True), ('MF', 2, True)])
def test_distributed_trainer_fit_matches_single_process(lfm, world, parallel):
    """CrossDomainTrainer.fit with config['dist_group'] over 2 ranks (SOURCE x2, TARGET, OVERLAP x2, TARGET again -- so the
    tables go dimension -> row -> dimension layout) and the sharded evaluation in the OVERLAP and TARGET phases, against the same
    trainer in one process: per-epoch losses, metrics, every table gathered back, the mapping."""
    import socket
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_dist_trainer_worker, args=(r, world, port, lfm, q, parallel)) for r in range(world)]
    for p in procs:
        p.start()
    res = _collect_ranks(q, procs)
    trainer, model, train, valid = _dist_trainer_setup(DEV, lfm)
    # the single-process run must see what the ranks saw: the ragged tails (< world rows) that the distributed run skips
    orig_step = model.fused_train_step
    # (a domain phase under parallel_domains splits its batches over HALF of the ranks)
    mod = lambda: world if (model.phase == 'OVERLAP' or not parallel) else world // 2
    model.fused_train_step = lambda inter, **kw: orig_step(type(inter)({k: v[:v.shape[0] - v.shape[0] % mod()] for k, v in inter.items()}), **kw)
    log = []
    orig = trainer._train_epoch
    trainer._train_epoch = lambda data, e: (log.append(orig(data, e)) or log[-1])
    trainer.fit(train, None, verbose=False, saved=False)
    final = trainer.evaluate(valid)
    model.set_phase('TARGET')
    score = trainer.evaluate(valid)['recall@5']
    model.set_phase('SOURCE')
    final = dict(final, source_recall=trainer.evaluate(trainer.valid_src)['recall@5'])
    for r in range(world):
        # epochs: SOURCE, SOURCE, TARGET, OVERLAP, OVERLAP, TARGET; under parallel_domains a rank only sees its own domain's
        mine = log if not parallel else [log[i] for i in ((0, 1, 3, 4) if r < world // 2 else (2, 3, 4, 5))]
        assert_close(torch.tensor(res[r][1]), torch.tensor(mine), rtol=5e-5, what=f'epoch losses rank{r}')
        assert abs(res[r][2] - score) < 1e-6 and res[r][3] == pytest.approx(final, abs=1e-6), (res[r][2], score, res[r][3], final)
        for k, v in res[r][4].items():
            assert_close(torch.from_numpy(v).to(DEV), getattr(model, k).weight.data, rtol=1e-4, atol=0.01 * 5e-2, what=k)
        for k, v in model.mapping.named_parameters():
            assert_close(torch.from_numpy(res[r][5][k]).to(DEV), v.detach(), rtol=1e-4, atol=0.01 * 5e-2, what=k)
        # the source user table stayed a row shard after OVERLAP (nothing trained it since); the target one went back to columns
        assert res[r][6] == [(len(range(r, model.total_num_users, world)), 16), 'dim']


def test_dim_layout_full_size_properties():
    """The dimension layout at the shape one rank sees at N = 8 (domain groups: 32 of 128 columns of the 50,000,001-user and
    20,000,001-item tables, its group's global batch of 8 x 1,048,576 triples), through properties that need no oracle run:
      * the partial scores are additive over a column cut: diff(32 columns) = diff(first 16) + diff(last 16), norms included --
        which is exactly what the all-reduce relies on;
      * partial_diff -> grad_from_diff -> sort -> applies moves the tables as the fused single-GPU step does on the same slice
        (same loss to 1e-6, same rows to fp32 rounding), i.e. cutting the step around the all-reduce changes nothing."""
    from recbole_cdr_amd import binding as B_
    from recbole_cdr_amd.dimshard import DimShardedBPRStep
    from recbole_cdr_amd.fused import FusedBPRStep
    free_b, _ = torch.cuda.mem_get_info()
    if free_b < 90e9:
        pytest.skip('needs ~80 GB of free HBM')
    nu, ni, Ds, Bg, TOI = 50_000_001, 20_000_001, 32, 8 << 20, 10_000_000
    g = torch.Generator(device=DEV); g.manual_seed(8)
    U = torch.empty(nu, Ds, device=DEV).normal_(0, 0.1, generator=g)
    I = torch.empty(ni, Ds, device=DEV).normal_(0, 0.1, generator=g)
    u = torch.randint(1, nu, (Bg,), device=DEV, generator=g)
    p = torch.randint(1, 1 + TOI, (Bg,), device=DEV, generator=g)
    n = torch.randint(1, 1 + TOI, (Bg,), device=DEV, generator=g)
    ctx, s = B_.ctx(U.device), B_.stream()
    diffs = []
    for lo, hi in ((0, 32), (0, 16), (16, 32)):
        Uc, Ic = (U if hi - lo == 32 else U[:, lo:hi].contiguous()), (I if hi - lo == 32 else I[:, lo:hi].contiguous())
        d = torch.empty(Bg + 2, device=DEV)
        B_.call('cdr_bpr_partial_diff', ctx, s, B_.f32(Uc), B_.f32(Ic), hi - lo, B_.i64(u), B_.i64(p), B_.i64(n), Bg, B_.f32(d))
        diffs.append(d)
        del Uc, Ic
    assert_close(diffs[1][:Bg] + diffs[2][:Bg], diffs[0][:Bg], rtol=1e-5, atol=1e-6, what='diff over a column cut')
    assert_close(diffs[1][Bg:] + diffs[2][Bg:], diffs[0][Bg:], rtol=1e-5, what='EmbLoss norms over a column cut')
    del diffs
    U2, I2 = U.clone(), I.clone()
    a = DimShardedBPRStep(U, I, Bg, opt='adam', lr=1e-3, reg_weight=0.01)            # no process group: the one-rank form of the step
    b = FusedBPRStep(U2, I2, Bg, opt='adam', lr=1e-3, reg_weight=0.01)
    la, lb = a.step(u, p, n).clone(), b.step(u, p, n).clone()
    assert_close(la[:6], lb[:6], rtol=1e-6, atol=0, what='loss, norms, coefficients')
    rows_u, rows_i = u[:4096], torch.cat([p[:2048], n[:2048]])
    assert_close(U[rows_u], U2[rows_u], rtol=1e-5, atol=1e-6, what='user rows')
    assert_close(I[rows_i], I2[rows_i], rtol=1e-5, atol=1e-6, what='item rows')
    fp = lambda t: t.view(-1)[::4099].double().sum()
    assert abs(float(fp(U) - fp(U2))) <= 1e-6 * abs(float(fp(U2))) + 1e-3 and abs(float(fp(I) - fp(I2))) <= 1e-6 * abs(float(fp(I2))) + 1e-3


def _bench_line_and_detail(p, tmp_path):
    """The LAST stdout line of a bench.py run (the bounded contract line) and the detail record it names (every leg)."""
    import json
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines[:5]
    assert len(lines[0]) <= 4096
    line = json.loads(lines[0])
    full = json.load(open(line['detail_file']))
    for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline', 'dtype', 'data'):
        assert line[k] == full[k], k
    return line, full


@pytest.mark.parametrize('world,extra', [(2, []), pytest.param(4, [], marks=pytest.mark.slow_gpu), pytest.param(8, [], marks=pytest.mark.slow_gpu), (2, ['--shard', 'dim'])])
def test_bench_multi_rank_line_contract(world, extra, tmp_path):
    """`bench.py --gpus N` as the driver launches it (torch.distributed.run, one rank per process) -- here with every rank on
    cuda:0 over gloo (CDR_BENCH_SHARED_GPU=1, small tables): stdout is exactly ONE JSON line from rank 0 with the contract's
    keys, the whole-job value, a roofline and an exchange object; the other legs (OVERLAP step, sharded full-sort) ran."""
    import json
    import os
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, CDR_BENCH_SHARED_GPU='1')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={world}', '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.join(root, 'bench.py'), '--gpus', str(world), '--steps', '3', '--warmup', '2',
           '--users', '400001', '--items-per-domain', '100000', '--batch', '8192', '--detail-file', str(tmp_path / 'detail.json')] + extra
    p = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    line, d = _bench_line_and_detail(p, tmp_path)
    assert d['n_gpus'] == world and d['steps'] == 3 and d['warmup'] == 2 and d['scaling'] == 'weak' and d['vs_baseline'] is None
    assert d['metric'] == 'training interactions/sec' and d['higher_is_better'] is True and d['dtype'] == 'f32'
    assert 'FUNCTIONAL CHECK ONLY' in d['data'] and 'sharding' in d['config'] and 'cpu_baseline' not in d
    B = d['config']['batch_per_domain_per_rank']
    assert abs(d['value'] - 2 * B * world / (d['ms_per_step'] * 1e-3)) / d['value'] < 1e-6
    r = d['roofline']
    assert r['bound'] == 'hbm' and abs(r['frac'] - r['achieved'] / r['peak']) < 1e-9
    assert d['exchange']['bytes_to_other_ranks_per_step_per_rank'] >= 0
    assert line['exchange']['bytes_to_other_ranks_per_step_per_rank'] == d['exchange']['bytes_to_other_ranks_per_step_per_rank']
    assert set(line['layouts']) == {'dim', 'row'} and line['roofline']['frac'] == d['roofline']['frac'] and 'cpu_baseline' not in line
    assert 0 < d['final_loss'] < 1 and d['overlap_phase']['loss'] >= 0
    assert d['fullsort']['U=1']['masked_top10']['ms'] > 0 and d['fullsort']['U=1024']['items_per_s'] > 0
    # BOTH layouts of the C5 tables in the one record: north_star's row shard and the dimension shard, each with its
    # own whole-job value, exchange bytes and sharding label; the headline fields are the first one's
    lay = d['layouts']
    assert set(lay) == {'dim', 'row'} and 'leg_errors' not in d, d.get('leg_errors')
    first = 'dim' if extra else 'row'                  # round 6: north_star's row shard carries the headline fields unless --shard dim
    assert d['north_star_layout'] == 'row' and (first in d['config']['sharding'].lower())
    assert lay[first]['value'] == d['value'] and lay[first]['ms_per_step'] == d['ms_per_step']
    for name, rec in lay.items():
        assert rec['n_gpus'] == world and rec['value'] > 0 and rec['scaling'] == 'weak' and name in rec['sharding'].lower(), (name, rec)
        assert rec['exchange']['bytes_to_other_ranks_per_step_per_rank'] >= 0
    if world > 1:
        assert lay['row']['exchange']['bytes_to_other_ranks_per_step_per_rank'] > lay['dim']['exchange']['bytes_to_other_ranks_per_step_per_rank']


@pytest.mark.slow_gpu
@pytest.mark.parametrize('inject,used', [('row:raise@1', 'dim'), ('dim,row:raise@0', 'replicas')])
def test_bench_multi_rank_layout_fallback(inject, used, tmp_path):
    """The first hardware run of `bench.py --gpus N` must not be losable: a layout that fails to come up on some
    rank is abandoned by every rank and the next one is tried (row -> dim since round 6), and when none comes up the ranks run independent replicas
    -- ONE JSON line with the contract's keys either way, `layout_fallback` saying what failed where and how many ranks each data group
    really has.  World 2 on cuda:0 over gloo with injected failures."""
    import json
    import os
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, CDR_BENCH_SHARED_GPU='1', CDR_PREFLIGHT_FAIL=inject)
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node=2', '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.join(root, 'bench.py'), '--gpus', '2', '--steps', '3', '--warmup', '2', '--users', '400001',
           '--items-per-domain', '100000', '--batch', '8192', '--preflight-seconds', '20', '--no-fullsort', '--detail-file', str(tmp_path / 'detail.json')]
    p = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    line, d = _bench_line_and_detail(p, tmp_path)
    fb = d['layout_fallback']
    assert fb['used'] == used and fb['fell_back'] is True and fb['attempts'][0]['layout'] == 'row' and fb['attempts'][0]['ok'] is False
    assert line['layout_fallback']['used'] == used and line['layout_fallback']['fell_back'] is True      # the line itself says what ran
    assert d['n_gpus'] == 2 and d['value'] > 0 and d['metric'] == 'training interactions/sec'
    B = d['config']['batch_per_domain_per_rank']
    assert abs(d['value'] - 2 * B * 2 / (d['ms_per_step'] * 1e-3)) / d['value'] < 1e-6
    if used == 'dim':
        assert fb['ranks_seen'] == {'source': 2, 'target': 2} and 'dim' in d['config']['sharding'].lower()
    else:
        assert 'INDEPENDENT REPLICAS' in d['config']['sharding'] and [a['ok'] for a in fb['attempts']] == [False, False]


def test_bench_comm_cabi_one_rank(tmp_path):
    """`bench.py --force-shard --comm cabi`: the row-sharded step's exchanges go through cdr_comm_init + cdr_a2a_ids /
    cdr_a2a_rows / cdr_allreduce_sum_f32; the line carries the communicator's own rank count.  One rank is what a one-GPU box allows."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, os.path.join(root, 'bench.py'), '--force-shard', '--comm', 'cabi', '--steps', '3', '--warmup', '2', '--users', '400001',
           '--items-per-domain', '100000', '--batch', '8192', '--no-fullsort', '--no-cpu-baseline', '--no-ingest',
           '--detail-file', str(tmp_path / 'detail.json')]
    p = subprocess.run(cmd, cwd=root, env=dict(os.environ, NCCL_SOCKET_IFNAME='lo'), capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    line, d = _bench_line_and_detail(p, tmp_path)
    assert line['layout_fallback']['comm'] == 'cabi' and line['layout_fallback']['ranks_seen'] == {'source': 1, 'target': 1}
    calls = d['cabi_calls_total']
    # per domain step (direct form): triples + request list (ids, one batch ahead: prefetched); rows + their squared norms + gradient rows
    assert calls['cdr_a2a_ids'] > 0 and calls['cdr_a2a_rows'] >= calls['cdr_a2a_ids'] and calls['cdr_allreduce_sum_f32'] > 0
    assert 'row' in d['config']['sharding'] and 'C ABI communicator' in d['config']['comm'] and 0 < d['final_loss'] < 10


def test_bench_comm_cabi_falls_back_to_torch_when_the_communicator_does_not_come_up(tmp_path):
    """Two ranks on ONE device: RCCL refuses the C ABI communicator (duplicate GPU), every rank abandons 'row-cabi' under the preflight
    watchdog and the row layout comes up over torch.distributed (gloo here) -- the line still prints and says so."""
    import os
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, CDR_BENCH_SHARED_GPU='1', NCCL_SOCKET_IFNAME='lo')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node=2', '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.join(root, 'bench.py'), '--gpus', '2', '--comm', 'cabi', '--steps', '3', '--warmup', '2', '--users', '400001',
           '--items-per-domain', '100000', '--batch', '8192', '--preflight-seconds', '30', '--no-fullsort',
           '--detail-file', str(tmp_path / 'detail.json')]
    p = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    line, d = _bench_line_and_detail(p, tmp_path)
    fb = d['layout_fallback']
    assert fb['attempts'][0]['layout'] == 'row-cabi' and fb['attempts'][0]['ok'] is False and fb['used'] == 'row' and fb['fell_back'] is True
    assert 'torch' in fb['comm'] and d['value'] > 0


@pytest.mark.parametrize('inject,used', [('', 'rowshard'), pytest.param('rowshard:raise@1', 'replica-dp', marks=pytest.mark.slow_gpu),
                                         pytest.param('rowshard,replica-dp:raise@0', 'replicas', marks=pytest.mark.slow_gpu)])
def test_bench_c4_multi_rank_layout_fallback(inject, used, tmp_path):
    """`bench.py --workload c4 --gpus 2` (BASELINE configs[3]: the row-sharded graph) under the same watchdog: rowshard -> replica data
    parallel -> independent replicas of the product's trainer loop; one JSON line either way.  cuda:0 shared over gloo."""
    import json
    import os
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, CDR_BENCH_SHARED_GPU='1', CDR_PREFLIGHT_FAIL=inject)
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node=2', '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.join(root, 'bench.py'), '--workload', 'c4', '--gpus', '2', '--steps', '4', '--warmup', '1',
           '--preflight-seconds', '25', '--no-cpu-baseline', '--detail-file', str(tmp_path / 'detail.json')]
    p = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    line, d = _bench_line_and_detail(p, tmp_path)
    fb = d['layout_fallback']
    assert fb['used'] == used and fb['fell_back'] == (used != 'rowshard') and d['n_gpus'] == 2 and d['value'] > 0 and 0 < d['final_loss'] < 10
    rows = d['config']['rows_per_step']
    want = rows * (1 if used == 'rowshard' else 2) / (d['ms_per_step'] * 1e-3)
    assert abs(d['value'] - want) / want < 1e-6
    if used == 'replicas':
        assert 'INDEPENDENT REPLICAS' in d['config']['workload'] and d['config']['via'] == 'CrossDomainTrainer.fit'


def _dist_ckpt_worker(rank, world, port, path, q):
    import os
    import faulthandler
    import torch.distributed as dist
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    faulthandler.dump_traceback_later(570, exit=True)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        import recbole_cdr_amd  # noqa: F401
        torch.cuda.set_device(0)
        S, T, O = 'SOURCE', 'TARGET', 'OVERLAP'
        tr_a, m_a, data_a, _ = _dist_trainer_setup(DEV, 'BPR', True, modes=[S, O, T, O], epochs=['1', '1', '1', '1'])
        tr_a.fit(data_a, None, verbose=False, saved=False)                       # uninterrupted
        tr_b, _m_b, data_b, _ = _dist_trainer_setup(DEV, 'BPR', True, modes=[S, O], epochs=['1', '1'])
        tr_b.fit(data_b, None, verbose=False, saved=False)
        tr_b.save_checkpoint(path, epoch=0)                                      # tables are row shards at this point (OVERLAP ran last)
        tr_c, m_c, data_c, _ = _dist_trainer_setup(DEV, 'BPR', True, modes=[T, O], epochs=['1', '1'])
        with torch.no_grad():
            for p in m_c.parameters():
                p.add_(1.0)                                                      # whatever the fresh model held must not matter
        tr_c.resume_checkpoint(path)
        tr_c.fit(data_c, None, verbose=False, saved=False)
        fa, fc = m_a.gather_full_tables(), m_c.gather_full_tables()
        same = {k: bool(torch.equal(fa[k], fc[k])) for k in fa}
        same.update({'mapping.' + k: bool(torch.equal(v, dict(m_c.mapping.named_parameters())[k])) for k, v in m_a.mapping.named_parameters()})
        faulthandler.cancel_dump_traceback_later()
        q.put((rank, same, os.path.exists(f'{path}.rank{rank}')))
    finally:
        dist.destroy_process_group()


def test_distributed_checkpoint_resume_continues_bit_exactly(tmp_path):
    """Sharded checkpoint (one file per rank: tables and moments in whatever layout they are in -- row shards after an OVERLAP
    phase --, update counts, the mapping and its Adam state): SOURCE, OVERLAP | save | fresh model, resume | TARGET, OVERLAP
    equals the uninterrupted four phases bit for bit on every table and on the mapping."""
    import socket
    import torch.multiprocessing as mp
    world = 2
    s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_dist_ckpt_worker, args=(r, world, port, str(tmp_path / 'ckpt.pth'), q)) for r in range(world)]
    for p in procs:
        p.start()
    res = _collect_ranks(q, procs)
    for r in range(world):
        assert res[r][2] and all(res[r][1].values()), res[r][1]


# ---------------------------------------------------------------------------------------------- row-sharded BiTGCF (configs[3])
def _bitgcf_shared_gpu_worker(rank, world, port, connect_way, q):
    import os
    import torch.distributed as dist
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        import recbole_cdr_amd  # noqa: F401
        from recbole_cdr_amd.bitgcf_shard import ShardedBiTGCF, NativeGraphOps
        from recbole_cdr_amd.trainer.trainer import DenseAdam
        from recbole_cdr_amd.data.synthetic import SyntheticCrossDomainDataset
        torch.cuda.set_device(0)
        ds, params, batches = _bitgcf_gpu_case()
        m = ShardedBiTGCF(ds.num_total_user, ds.num_total_item, ds.num_overlap_user, ds.num_overlap_item, ds.s_pairs, ds.t_pairs, 64, 2,
                          0.8, 0.7, connect_way, 0.001, NativeGraphOps(DEV), init=params)
        opt = DenseAdam(list(m.params.values()), lr=0.01)
        losses = []
        for b in batches:
            opt.zero_grad(set_to_none=True)
            ls, lt = m.loss_and_grads(b)
            losses.append((float(ls), float(lt)))
            opt.step()
        full = m.full_tables()
        q.put((rank, losses, {k: v.cpu().numpy() for k, v in full.items()}, [t.cpu().numpy() for t in m.propagated_tables()]))
    finally:
        dist.destroy_process_group()


def _bitgcf_gpu_case():
    from recbole_cdr_amd.data.synthetic import SyntheticCrossDomainDataset
    ds = SyntheticCrossDomainDataset(OU=301, TOU=200, SOU=150, OI=1, TOI=400, SOI=350, n_source_inter=4000, n_target_inter=5000, seed=5)
    g = torch.Generator().manual_seed(9)
    params = {k: torch.randn(ds.num_total_user if '_user_' in k else ds.num_total_item, 64, generator=g) * 0.1
              for k in ('source_user_embedding.weight', 'source_item_embedding.weight', 'target_user_embedding.weight',
                        'target_item_embedding.weight')}
    rng = np.random.RandomState(2)
    batches = [dict(ds.pointwise_batch('source', 128, 1, rng, DEV), **ds.pointwise_batch('target', 128, 1, rng, DEV)) for _ in range(3)]
    return ds, params, batches


@pytest.mark.parametrize('world,connect_way', [(2, 'concat'), (3, 'mean')])
def test_row_sharded_bitgcf_native_ranks_share_one_gpu(world, connect_way):
    """bitgcf_shard.ShardedBiTGCF on the NATIVE kernels (cdr_graph_layer_fwd_rows / _bwd_rows, transfer, normalise, gather-dot-BCE),
    `world` ranks on cuda:0 over gloo: three Adam steps -- both losses every step, the four tables gathered back and the propagated
    tables -- equal the single-GPU BiTGCF model (dense autograd + DenseAdam) on the same batches."""
    import socket
    import torch.multiprocessing as mp
    from recbole_cdr_amd.model.cross_domain_recommender.bitgcf import BiTGCF
    from recbole_cdr_amd.trainer.trainer import DenseAdam
    s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_bitgcf_shared_gpu_worker, args=(r, world, port, connect_way, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = _collect_ranks(q, procs)
    ds, params, batches = _bitgcf_gpu_case()
    cfg = base_config(DEV, embedding_size=64, n_layers=2, reg_weight=0.001, lambda_source=0.8, lambda_target=0.7, drop_rate=0.0,
                      connect_way=connect_way)
    ref = BiTGCF(cfg, ds).to(DEV)
    load_params(ref, params)
    opt = DenseAdam(ref.parameters(), lr=0.01)
    want = []
    for b in batches:
        opt.zero_grad(set_to_none=True)
        ls, lt = ref.calculate_loss(b)
        (ls + lt).sum().backward()
        want.append((float(ls), float(lt)))
        opt.step()
    with torch.no_grad():
        prop = ref.forward()
    for r, losses, full, got_prop in res:
        for (a, b), (c, d) in zip(losses, want):
            assert abs(a - c) <= 1e-5 * abs(c) and abs(b - d) <= 1e-5 * abs(d), (r, losses, want)
        for k, v in ref.named_parameters():
            assert_close(torch.from_numpy(full[k]), v.detach(), rtol=1e-5, atol=0.01 * 1e-2, what=k)
        for a, b in zip(got_prop, prop):
            # three Adam steps in: the tables agree within Adam's drift bound (1e-2 of one update), and so do the propagated rows
            assert_close(torch.from_numpy(a), b, rtol=1e-5, atol=0.01 * 1e-2, what='propagated table')


# ---------------------------------------------------------------------------------------------- C-ABI exchanges (SURVEY 8b family (10))
def test_comm_family_one_rank_communicator():
    """cdr_comm_* / cdr_a2a_* / cdr_allgather_scores / cdr_allreduce_sum_f32 through the C ABI inside a torch process (RCCL bound at run time to
    the librccl torch loaded): a ONE-rank communicator is all a one-GPU box allows (RCCL refuses two ranks on one device), so this checks the
    binding, the unique-id hand-over, the stream ordering and the count / offset arithmetic -- every exchange is then the identity."""
    import ctypes
    from recbole_cdr_amd import binding as B_
    lib = B_.load()
    idb = ctypes.create_string_buffer(128)
    B_.call('cdr_comm_unique_id', ctypes.cast(idb, ctypes.c_void_p))
    assert any(idb.raw)
    comm = ctypes.c_void_p()
    torch.cuda.set_device(0)
    B_.call('cdr_comm_init', ctypes.cast(ctypes.pointer(comm), ctypes.c_void_p), 0, 1, ctypes.cast(idb, ctypes.c_void_p))
    assert comm.value
    try:
        r, w = ctypes.c_int(-1), ctypes.c_int(-1)
        B_.call('cdr_comm_info', comm, ctypes.cast(ctypes.pointer(r), ctypes.c_void_p), ctypes.cast(ctypes.pointer(w), ctypes.c_void_p))
        assert (r.value, w.value) == (0, 1)
        g = torch.Generator().manual_seed(0)
        ids = torch.randint(0, 1 << 40, (1000,), generator=g).to(DEV)
        got = torch.empty_like(ids)
        cnt = (ctypes.c_int64 * 1)(1000)
        B_.call('cdr_a2a_ids', comm, B_.stream(), B_.i64(ids), ctypes.cast(cnt, ctypes.c_void_p), B_.i64(got), ctypes.cast(cnt, ctypes.c_void_p))
        rows = torch.randn(333, 64, generator=g).to(DEV)
        back = torch.empty_like(rows)
        rc = (ctypes.c_int64 * 1)(333)
        B_.call('cdr_a2a_rows', comm, B_.stream(), B_.f32(rows), ctypes.cast(rc, ctypes.c_void_p), B_.f32(back), ctypes.cast(rc, ctypes.c_void_p), 64)
        sc = torch.randn(5000, generator=g).to(DEV)
        allsc = torch.empty_like(sc)
        B_.call('cdr_allgather_scores', comm, B_.stream(), B_.f32(sc), sc.numel(), B_.f32(allsc))
        red = sc.clone()
        B_.call('cdr_allreduce_sum_f32', comm, B_.stream(), B_.f32(red), red.numel())
        torch.cuda.synchronize()
        assert torch.equal(got, ids) and torch.equal(back, rows) and torch.equal(allsc, sc) and torch.equal(red, sc)
        # a negative count is refused before anything is enqueued
        bad = (ctypes.c_int64 * 1)(-1)
        assert lib.cdr_a2a_ids(comm, B_.stream(), B_.i64(ids), ctypes.cast(bad, ctypes.c_void_p), B_.i64(got), ctypes.cast(cnt, ctypes.c_void_p)) != 0
    finally:
        B_.call('cdr_comm_destroy', comm)
