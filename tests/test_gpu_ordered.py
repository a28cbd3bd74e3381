"""GPU (-m gpu): the ordered dense backward (cdr_ordered_bwd, csrc/cdr_ordered.hip) -- the drop-in losses' dense gradients as
occurrence-order sums out of one launch: what ``set_deterministic(True)`` runs at the reference's batch sizes.

Three routes to the same gradient exist and are held against each other here:
  atomic   the default: fp32 atomics, torch's embedding backward in spirit (emcdr.py:123-154 under autograd)
  sorted   ``set_deterministic(True)`` + ``set_ordered_backward(False)``: gathers, per-occurrence rows, id sort, segmented scatter
  ordered  ``set_deterministic(True)``, lists within ``ordered_max()``: one launch
ordered vs sorted: the same sum in the same order (1e-6; bit-equal except where the compiler contracts the term differently);
ordered vs atomic: 1e-5 (the order of the adds differs); ordered twice: bit-equal.  Tolerances are north_star's 1e-5 relative."""
import pytest
import torch

from helpers import DEV

pytestmark = pytest.mark.gpu


def _routes(fn):
    from recbole_cdr_amd import functional as F_
    try:
        F_.set_ordered_backward(False)
        F_.set_deterministic(False)
        atomic = fn()
        F_.set_deterministic(True)
        sorted_ = fn()
        F_.set_ordered_backward(True, max_entries=16384)
        o1, o2 = fn(), fn()
    finally:
        F_.set_deterministic(False)
        F_.set_ordered_backward(True)
    return atomic, sorted_, o1, o2


def _hold(name, atomic, sorted_, o1, o2):
    for k, (a, s, x, y) in enumerate(zip(atomic, sorted_, o1, o2)):
        assert torch.equal(x, y), f'{name}[{k}]: two ordered runs differ'
        scale = float(a.abs().max()) + 1e-30
        torch.testing.assert_close(x, s, rtol=1e-6, atol=1e-6 * scale, msg=f'{name}[{k}] ordered vs sorted')
        torch.testing.assert_close(x, a, rtol=1e-5, atol=1e-5 * scale, msg=f'{name}[{k}] ordered vs atomic')


@pytest.mark.parametrize('nu,ni,D,n,hot', [(37, 29, 16, 600, 0.0), (5000, 3000, 64, 2048, 0.0), (5000, 3000, 64, 2048, 0.4), (900, 700, 128, 4096, 0.1),
                                           (50, 40, 48, 333, 0.0), (20000, 20000, 64, 8000, 0.02), (11, 7, 192, 257, 0.0)])
def test_ordered_backward_of_every_drop_in_loss(nu, ni, D, n, hot):
    """Every autograd node that scatters into a dense table gradient, on batches full of repeated ids (and, with ``hot``, one user and
    one item that take that share of the batch: chains far longer than the eight occurrences a group gathers per round)."""
    from recbole_cdr_amd import functional as F_, binding as B_
    gen = torch.Generator().manual_seed(nu * 131 + n)
    U0, I0 = torch.randn(nu, D, generator=gen) * 0.3, torch.randn(ni, D, generator=gen) * 0.3

    def ids(hi):
        t = torch.randint(0, hi, (n,), generator=gen)
        if hot:
            t[torch.rand(n, generator=gen) < hot] = hi // 3
        return t.to(DEV)

    u, p_, q_ = ids(nu), ids(ni), ids(ni)
    y = (torch.rand(n, generator=gen) < 0.5).float().to(DEV)
    wrow = torch.arange(n, device=DEV).view(-1, 1).float().sin()

    def leaves(*ts):
        return [t.clone().to(DEV).requires_grad_(True) for t in ts]

    def bpr():
        U, I = leaves(U0, I0)
        (F_.BPRGatherLoss.apply(U, I, u, p_, q_, 1e-10, 0.01) * 1.7).sum().backward()
        return [U.grad, I.grad]

    def point(kind, reg):
        U, I = leaves(U0, I0)
        F_.PointGatherLoss.apply(kind, U, I, None, None, u, p_, y, reg)[0].sum().backward()
        return [U.grad, I.grad]

    def point_sep():
        U, I, RU, RI = leaves(U0, I0, U0.flip(0) * 0.5, I0.flip(0) * 0.5)
        (F_.PointGatherLoss.apply(B_.CDR_LOSS_BCE, U, I, RU, RI, u, p_, y, 0.05)[0] * 0.8).sum().backward()
        return [U.grad, I.grad, RU.grad, RI.grad]

    def point_shared():
        S, = leaves(torch.cat([U0, I0]))
        F_.PointGatherLoss.apply(B_.CDR_LOSS_BCE, S, S, None, None, u, p_ + nu, y, 0.0)[0].sum().backward()
        return [S.grad]

    def pair():
        U, I = leaves(U0, I0)
        F_.TwoDomainPointLoss.apply(B_.CDR_LOSS_BCE, U, I, u, p_, y, 0.01, u.flip(0), q_, 1 - y, 0.03, 0.3)[0].sum().backward()
        return [U.grad, I.grad]

    def gather():
        W, = leaves(U0)
        (F_.gather_rows(W, u) * wrow).sum().backward()
        return [W.grad]

    def two_stack():
        S, T = leaves(torch.cat([U0, I0]), torch.cat([U0.flip(0), I0.flip(0)]))
        ls, lt = F_.TwoStackPointLoss.apply(B_.CDR_LOSS_BCE, S, T, nu, u, p_, y, u.flip(0), q_, 1 - y)
        (0.7 * ls + 1.3 * lt).sum().backward()
        return [S.grad, T.grad]

    def embloss_rows():
        U, I = leaves(U0, I0)
        (F_.EmbLossRows.apply(U, I, u, p_) * 0.9).sum().backward()
        return [U.grad, I.grad]

    cases = {'bpr': bpr, 'mse': lambda: point(B_.CDR_LOSS_MSE, 0.02), 'bce': lambda: point(B_.CDR_LOSS_BCE, 0.0), 'sep': point_sep,
             'shared': point_shared, 'pair': pair, 'gather': gather, 'two_stack': two_stack, 'embloss_rows': embloss_rows}
    for name, fn in cases.items():
        _hold(f'{name} nu={nu} D={D} n={n} hot={hot}', *_routes(fn))


def test_ordered_backward_is_what_set_deterministic_runs_and_where_it_stops():
    """At the reference's batch (2,048 triples: lists of 2,048 and 4,096) ``set_deterministic(True)`` IS the one ordered launch -- checked
    by counting the library calls --, a list beyond ``ordered_max()`` keeps the sorted form, and without set_deterministic nothing
    changes (the atomic scatter)."""
    from recbole_cdr_amd import functional as F_, binding as B_
    gen = torch.Generator().manual_seed(3)
    D = 64
    U0, I0 = torch.randn(7000, D, generator=gen) * 0.1, torch.randn(4000, D, generator=gen) * 0.1
    seen = []
    real = B_.call

    def spy(name, *a):
        seen.append(name)
        return real(name, *a)

    def run(n):
        U, I = U0.clone().to(DEV).requires_grad_(True), I0.clone().to(DEV).requires_grad_(True)
        u, p_, q_ = (torch.randint(0, h, (n,), generator=gen).to(DEV) for h in (7000, 4000, 4000))
        loss = F_.BPRGatherLoss.apply(U, I, u, p_, q_, 1e-10, 0.01)
        del seen[:]
        B_.call = spy
        F_.B_.call = spy
        try:
            loss.sum().backward()
        finally:
            B_.call = real
            F_.B_.call = real
        return list(seen)

    assert run(2048) == ['cdr_bpr_bwd_dense']                                               # the default: atomics
    assert F_.ordered_max() == 8192
    try:
        F_.set_deterministic(True)
        assert run(2048) == ['cdr_ordered_bwd'] and run(4096) == ['cdr_ordered_bwd']
        got = run(8192)                                                                     # item list 16,384 > ordered_max()
        assert 'cdr_ordered_bwd' not in got and 'cdr_scatter_rows_sorted' in got
        F_.set_ordered_backward(True, max_entries=16384)
        assert run(8192) == ['cdr_ordered_bwd']                                             # raised: up to the kernel's own limit
        assert 'cdr_scatter_rows_sorted' in run(20000)
    finally:
        F_.set_deterministic(False)
        F_.set_ordered_backward(True)


def test_ordered_bwd_c_abi_terms_wide_ids_and_accumulate():
    """cdr_ordered_bwd through the C ABI on hand-made lists, against a float64 loop on the host: two segments on one buffer, the
    occurrence index as the source row (xid NULL), a subtraction term, the accumulate form -- and ids that differ only ABOVE bit 32
    (the LDS test compares low words; the hit has to be confirmed on 64 bits) in a table of more than 2^32 rows."""
    from recbole_cdr_amd import binding as B_
    gen = torch.Generator().manual_seed(9)
    D, rows, n = 32, 40, 300
    X = torch.randn(n, D, generator=gen).to(DEV)
    Y = torch.randn(50, D, generator=gen).to(DEV)
    R = torch.randn(rows, D, generator=gen).to(DEV)
    ids_a = torch.randint(0, rows, (n,), generator=gen).to(DEV)
    ids_b = torch.randint(0, rows, (n // 2,), generator=gen).to(DEV)
    yid = torch.randint(0, 50, (n // 2,), generator=gen).to(DEV)
    xid = torch.randint(0, n, (n // 2,), generator=gen).to(DEV)
    coef = torch.randn(n // 2, generator=gen).to(DEV)
    go = torch.tensor([1.3], device=DEV)
    norm = torch.tensor([2.5], device=DEV)
    g = torch.full((rows, D), 0.25, device=DEV)

    def seg(ids, **kw):
        sg = B_.OrdSeg()
        sg.ids, sg.n, sg.sign, sg.go_scale = ids.data_ptr(), ids.numel(), 1.0, 1.0
        for k, v in kw.items():
            setattr(sg, k, v)
        return sg

    def launch(gptr, segs, accumulate, D_=D):
        arr = (B_.OrdList * 1)()
        arr[0].g, arr[0].g_stride, arr[0].nseg, arr[0].accumulate = gptr, D_, len(segs), accumulate
        for k, s in enumerate(segs):
            arr[0].seg[k] = s
        B_.call('cdr_ordered_bwd', B_.stream(), D_, arr, 1)

    s_a = seg(ids_a, X=X.data_ptr(), x_stride=D)                                                   # plain rows by occurrence index
    s_b = seg(ids_b, coef=coef.data_ptr(), sign=-1.0, go=go.data_ptr(), go_scale=0.5, X=X.data_ptr(), xid=xid.data_ptr(),
              Y=Y.data_ptr(), yid=yid.data_ptr(), x_stride=D, R=R.data_ptr(), r_stride=D, norm=norm.data_ptr(), reg_weight=0.7, B=n // 2)
    launch(g.data_ptr(), [s_a, s_b], 1)
    want = torch.full((rows, D), 0.25, dtype=torch.float64)
    add = torch.zeros(rows, D, dtype=torch.float64)
    Xc, Yc, Rc = X.double().cpu(), Y.double().cpu(), R.double().cpu()
    for j, r in enumerate(ids_a.tolist()):
        add[r] += Xc[j]
    gs = 1.3 * 0.5
    c = gs * 0.7 / ((n // 2) * 2.5)
    for j, r in enumerate(ids_b.tolist()):
        add[r] += -1.0 * (gs * float(coef[j])) * (Xc[int(xid[j])] - Yc[int(yid[j])]) + c * Rc[r]
    torch.testing.assert_close(g.double().cpu(), want + add, rtol=1e-5, atol=1e-5)
    untouched = torch.ones(rows, dtype=torch.bool)
    untouched[ids_a.cpu()] = False
    untouched[ids_b.cpu()] = False
    assert bool((g.cpu()[untouched] == 0.25).all())

    # ids that differ only above bit 32 (a table of more than 2^32 rows of 4 floats: 68.7 GB of the 288): the LDS test compares low
    # words, the owner and the later occurrences have to be confirmed on 64 bits
    D2, hi = 4, 1 << 32
    try:
        big = torch.empty((hi + 16) * D2, device=DEV, dtype=torch.float32)
    except RuntimeError:
        pytest.skip('no room for a 68.7 GB table on this device')
    big[:16 * D2] = 0.0
    big[hi * D2:] = 0.0
    wide = torch.tensor([5, 5 + hi, 5, 5 + hi, 6, 5 + hi], dtype=torch.int64, device=DEV)
    src = (torch.arange(6 * D2, dtype=torch.float32).view(6, D2) + 1.0).to(DEV)
    launch(big.data_ptr(), [seg(wide, X=src.data_ptr(), x_stride=D2)], 0, D_=D2)
    torch.cuda.synchronize()
    lo_rows, hi_rows = big[:16 * D2].view(16, D2).cpu(), big[hi * D2:].view(16, D2).cpu()
    s_ = src.cpu()
    assert torch.equal(lo_rows[5], s_[0] + s_[2]) and torch.equal(lo_rows[6], s_[4]) and torch.equal(hi_rows[5], (s_[1] + s_[3]) + s_[5])
    assert float(lo_rows[:5].abs().max()) == 0.0 and float(hi_rows[6:].abs().max()) == 0.0
    del big


@pytest.mark.parametrize('connect_way,L,D,reg', [('concat', 2, 64, 0.01), ('mean', 1, 32, 0.0)])
def test_bitgcf_one_node_loss_under_set_deterministic(connect_way, L, D, reg):
    """BiTGCF.calculate_loss under ``set_deterministic(True)``: the one-node form stays (its stack scatter and the EmbLoss rows ADDED into
    the propagation's table gradients go through the ordered launch, the latter with ``accumulate``): every table gradient bit-equal from
    run to run, 1e-5 from the default (atomic) run, losses 1e-6; batches with repeated users and items (bitgcf.py:207-247)."""
    import numpy as np
    from helpers import FakeDataset, base_config, to_dev, assert_close
    from oracle.common import IdSpace
    from recbole_cdr_amd import functional as F_, binding as B_
    from recbole_cdr_amd.model.cross_domain_recommender.bitgcf import BiTGCF
    from recbole_cdr_amd.utils import total_loss
    ids = IdSpace(OU=30, TOU=25, SOU=20, OI=1, TOI=40, SOI=35)
    rng = np.random.RandomState(5)
    su_ = np.r_[1:ids.OU, ids.OU + ids.TOU:ids.total_num_users]; si_ = np.r_[ids.OI + ids.TOI:ids.total_num_items]
    tu_ = np.r_[1:ids.OU + ids.TOU]; ti_ = np.r_[1:ids.OI + ids.TOI]
    s_pairs = np.unique(np.stack([rng.choice(su_, 400), rng.choice(si_, 400)], 1), axis=0)
    t_pairs = np.unique(np.stack([rng.choice(tu_, 500), rng.choice(ti_, 500)], 1), axis=0)
    ds = FakeDataset(ids, s_pairs=s_pairs.astype(np.int64), t_pairs=t_pairs.astype(np.int64))
    B = 200
    inter = {'source_user_id': torch.from_numpy(rng.choice(su_, B)), 'source_item_id': torch.from_numpy(rng.choice(si_, B)),
             'source_label': torch.from_numpy((rng.rand(B) < 0.5).astype(np.float32)),
             'target_user_id': torch.from_numpy(rng.choice(tu_, B)), 'target_item_id': torch.from_numpy(rng.choice(ti_, B)),
             'target_label': torch.from_numpy((rng.rand(B) < 0.5).astype(np.float32))}
    inter['source_user_id'][:40] = inter['source_user_id'][0]; inter['target_item_id'][:25] = inter['target_item_id'][0]
    seen = []
    real = B_.call

    def spy(name, *a):
        seen.append(name)
        return real(name, *a)

    def run():
        cfg = base_config(DEV, embedding_size=D, n_layers=L, reg_weight=reg, lambda_source=0.8, lambda_target=0.7, drop_rate=0.0,
                          connect_way=connect_way)
        torch.manual_seed(1)
        model = BiTGCF(cfg, ds).to(DEV)
        model.train()
        assert model.fused_loss
        F_.B_.call = spy
        try:
            losses = model.calculate_loss(to_dev(inter, DEV))
            total_loss(losses).sum().backward()
        finally:
            F_.B_.call = real
        return torch.stack([x.detach().reshape(()) for x in losses]), {k: v.grad.clone() for k, v in model.named_parameters()}

    want = run()
    assert 'cdr_ordered_bwd' not in seen and 'cdr_point_bwd_dense_pair' in seen
    try:
        F_.set_deterministic(True)
        del seen[:]
        a, b = run(), run()
    finally:
        F_.set_deterministic(False)
    assert 'cdr_point_fwd_pair_ex' in seen and 'cdr_point_bwd_dense_pair' not in seen and 'cdr_embloss_bwd_dense_pair' not in seen
    assert seen.count('cdr_ordered_bwd') == (4 if reg else 2)
    assert_close(a[0], want[0], rtol=1e-6, what='losses')
    for k in want[1]:
        assert torch.equal(a[1][k], b[1][k]), k
        assert_close(a[1][k], want[1][k], rtol=1e-5, what=f'grad {k}')
