"""The product's trainer loop on the captured step: the device batch producer (csrc/cdr_sampler.hip
batch_produce_kernel) against the loader it replaces, and ``CrossDomainTrainer.fit`` with one hipGraph replay per batch against the
same steps run eagerly.  Reference: recbole_cdr/trainer/trainer.py:43-76, data/dataloader.py:25-186, crossdomain_sampler.py:139-175."""
import ctypes

import numpy as np
import pytest
import torch

from helpers import DEV, FakeDataset, base_config, assert_close

pytestmark = pytest.mark.gpu


def _dataset(seed=0, n_s=700, n_t=500):
    from oracle.common import IdSpace
    ids = IdSpace(OU=40, TOU=30, SOU=35, OI=1, TOI=60, SOI=70)
    rng = np.random.RandomState(seed)
    src_u = np.array(list(range(1, ids.OU)) + list(range(ids.OU + ids.TOU, ids.total_num_users)))
    src_i = np.arange(ids.OI + ids.TOI, ids.total_num_items)
    tgt_u, tgt_i = np.arange(1, ids.OU + ids.TOU), np.arange(1, ids.OI + ids.TOI)
    s_pairs = np.unique(np.stack([rng.choice(src_u, n_s), rng.choice(src_i, n_s)], 1), axis=0)
    t_pairs = np.unique(np.stack([rng.choice(tgt_u, n_t), rng.choice(tgt_i, n_t)], 1), axis=0)
    rng.shuffle(s_pairs); rng.shuffle(t_pairs)
    ds = FakeDataset(ids, s_pairs, t_pairs)
    return ids, ds, s_pairs, t_pairs


def _loaders(ids, ds, s_pairs, t_pairs, input_type, batch, k, shuffle=False, ob=16, seed=5):
    from recbole_cdr_amd.data import CrossDomainDataloader, OverlapDataloader, DomainTrainLoader
    from recbole_cdr_amd.sampler import DeviceNegSampler
    dt = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(DEV)
    s_smp, t_smp = DeviceNegSampler(ds, 'source', s_pairs, DEV, seed=seed), DeviceNegSampler(ds, 'target', t_pairs, DEV, seed=seed + 1)
    gen = lambda j: torch.Generator(device=DEV).manual_seed(1000 * seed + j) if shuffle else None
    return CrossDomainDataloader(
        DomainTrainLoader({'source_user_id': dt(s_pairs[:, 0]), 'source_item_id': dt(s_pairs[:, 1])}, 'source_user_id', 'source_item_id',
                          'source_label', 'neg_', batch, k, input_type, s_smp, shuffle=shuffle, generator=gen(1)),
        DomainTrainLoader({'target_user_id': dt(t_pairs[:, 0]), 'target_item_id': dt(t_pairs[:, 1])}, 'target_user_id', 'target_item_id',
                          'target_label', 'neg_', batch, k, input_type, t_smp, shuffle=shuffle, generator=gen(2)),
        OverlapDataloader(ids.OU, ob, device=DEV, shuffle=shuffle, generator=gen(3)))


@pytest.mark.parametrize('pointwise,k', [(False, 1), (False, 3), (True, 1), (True, 4)])
def test_batch_producer_layout_draws_and_cursor(pointwise, k):
    """One launch = the loader's batch: rows [start, start + S) tiled in recbole's layout, negatives bit-equal to
    cdr_neg_sample_uniform with the producer's seed rule, never an interacted item; the device cursor moves on by S per launch."""
    from recbole_cdr_amd import binding as B_
    from recbole_cdr_amd.data.producer import DeviceBatchProducer
    from recbole_cdr_amd.utils import InputType
    ids, ds, s_pairs, t_pairs = _dataset()
    it = InputType.POINTWISE if pointwise else InputType.PAIRWISE
    batch = 64 * (1 + k if pointwise else k)
    dl = _loaders(ids, ds, s_pairs, t_pairs, it, batch, k).source_dataloader
    assert DeviceBatchProducer.supports(dl)
    prod = DeviceBatchProducer(dl)
    S = dl.step
    assert S == 64
    smp = dl.neg_sampler
    used = set(map(tuple, s_pairs.tolist()))
    for call in range(3):
        prod.launch()
        torch.cuda.synchronize()
        start = call * S
        u, i = s_pairs[start:start + S, 0], s_pairs[start:start + S, 1]
        T = 1 + k if pointwise else k
        got_u = prod.fields['source_user_id'].cpu().numpy()
        assert (got_u == np.tile(u, T)).all()
        users_d = torch.from_numpy(u.copy()).to(DEV)
        want_neg = torch.empty(S * k, device=DEV, dtype=torch.int64)
        seed = (smp.graph_seed() + call * 0x85EBCA77C2B2AE63) & 0xFFFFFFFFFFFFFFFF
        lo0, hi0, lo1, hi1 = smp.ranges
        B_.call('cdr_neg_sample_uniform', B_.stream(), B_.i64(users_d), S, k, lo0, hi0, lo1, hi1, B_.i64(smp.indptr), B_.i64(smp.indices),
                seed, B_.i64(want_neg), B_.raw(smp.fail))
        want_neg = want_neg.cpu().numpy()
        items = prod.fields['source_item_id'].cpu().numpy()
        if pointwise:
            assert (items[:S] == i).all() and (items[S:] == want_neg).all()
            lab = prod.fields['source_label'].cpu().numpy()
            assert (lab[:S] == 1).all() and (lab[S:] == 0).all()
            neg = items[S:]
        else:
            assert (items == np.tile(i, k)).all()
            neg = prod.fields['neg_source_item_id'].cpu().numpy()
            assert (neg == want_neg).all()
            assert prod.fields.k_major == k
        assert all((int(a), int(b)) not in used for a, b in zip(np.tile(u, k), neg))
        assert ((neg >= ids.OI + ids.TOI) & (neg < ids.total_num_items)).all()          # source candidates (OI = 1: no overlapped items)
        cur = prod.cursor.cpu().numpy()
        assert cur[0] == (call + 1) * S and cur[1] == call + 1 and cur[2] == 0
    assert int(smp.fail.item()) == 0


@pytest.mark.parametrize('pointwise,k,two', [(False, 1, False), (True, 2, True)])
def test_adam_launch_that_also_produces_the_next_batch_equals_the_two_launches(pointwise, k, two):
    """Round 6 (cdr_adam_multi_dev_produce, trainer.DenseAdam.produce_jobs): the dense Adam of a step with the loader's next batch produced
    in workgroups behind its own -- parameters, moments, update counts, the produced batch (both loaders of a BOTH state), the cursors and
    the draw counters equal the optimizer launch followed by the producer launch, bit for bit, over several steps."""
    from recbole_cdr_amd.data.producer import DeviceBatchProducer, CompositeProducer
    from recbole_cdr_amd.trainer.trainer import DenseAdam
    from recbole_cdr_amd.utils import InputType
    ids, ds, s_pairs, t_pairs = _dataset()
    it = InputType.POINTWISE if pointwise else InputType.PAIRWISE
    batch = 64 * (1 + k if pointwise else k)

    def setup():
        dls = _loaders(ids, ds, s_pairs, t_pairs, it, batch, k)
        parts = [DeviceBatchProducer(dls.source_dataloader)] + ([DeviceBatchProducer(dls.target_dataloader)] if two else [])
        prod = CompositeProducer(parts) if two else parts[0]
        torch.manual_seed(4)
        ps = [torch.nn.Parameter(torch.randn(300, 64, device=DEV)), torch.nn.Parameter(torch.randn(17, device=DEV)), torch.nn.Parameter(torch.randn(40, 33, device=DEV))]
        return prod, ps, DenseAdam(ps, lr=0.01)
    pa, psa, oa = setup()
    pb, psb, ob = setup()
    g = torch.Generator(device=DEV); g.manual_seed(9)
    for step in range(4):
        grads = [torch.randn(p.shape, device=DEV, generator=g) for p in psa]
        for p, q, gr in zip(psa, psb, grads):
            p.grad, q.grad = gr.clone(), gr.clone()
        oa.step(); pa.launch()                                   # two launches
        ob.produce_jobs = pb.jobs(); ob.step()                   # one
        assert ob.produce_jobs is None
        torch.cuda.synchronize()
        for p, q in zip(psa, psb):
            assert torch.equal(p, q) and torch.equal(oa.state[p]['exp_avg'], ob.state[q]['exp_avg']) and torch.equal(oa.state[p]['exp_avg_sq'], ob.state[q]['exp_avg_sq'])
            assert int(oa.state[p]['step']) == int(ob.state[q]['step']) == step + 1
        for k_ in pa.fields:
            assert torch.equal(pa.fields[k_], pb.fields[k_]), k_
        for ca, cb in zip(pa.state_tensors(), pb.state_tensors()):
            assert torch.equal(ca, cb) and int(ca[0]) == (step + 1) * 64 and int(ca[2]) == 0
    # a step in which no parameter has a gradient still produces its batch
    for q in psb:
        q.grad = None
    ob.produce_jobs = pb.jobs(); ob.step(); pa.launch()
    torch.cuda.synchronize()
    for k_ in pa.fields:
        assert torch.equal(pa.fields[k_], pb.fields[k_]), k_


def test_batch_producer_overlap_slices_and_graph_replay():
    """k = 0: OverlapDataloader's [OB, 1] slices; captured once, every replay yields the next slice."""
    from recbole_cdr_amd import binding as B_
    from recbole_cdr_amd.data.producer import DeviceBatchProducer
    from recbole_cdr_amd.utils import InputType
    ids, ds, s_pairs, t_pairs = _dataset()
    ov = _loaders(ids, ds, s_pairs, t_pairs, InputType.PAIRWISE, 32, 1, ob=8).overlap_dataloader
    prod = DeviceBatchProducer(ov)
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        prod.launch()
    torch.cuda.synchronize()
    prod.resync()
    g = torch.cuda.CUDAGraph()
    with B_.capturing(g, side):
        prod.launch()
    prod.resync()
    for b in range(ids.OU // 8):
        g.replay()
        torch.cuda.synchronize()
        assert prod.fields['overlap'].shape == (8, 1)
        assert prod.fields['overlap'].view(-1).tolist() == list(range(8 * b, 8 * b + 8))
        prod.advance()
    assert ov.pr == ids.OU // 8 * 8 and not prod.full_ahead() or ids.OU % 8 == 0


def _params(model):
    return {k: v.detach().clone() for k, v in model.named_parameters()}


@pytest.mark.parametrize('mapping,graph_pipeline', [('linear', True), ('non_linear', True)])
def test_trainer_fit_on_replays_equals_the_same_steps_run_eagerly(mapping, graph_pipeline):
    """CrossDomainTrainer.fit(SOURCE, TARGET, OVERLAP) on device loaders: every full batch is one hipGraph replay that produces the
    batch itself.  Reference run: the SAME launches issued eagerly (producer launch -> zero_grad -> calculate_loss -> backward ->
    DenseAdam.step; ragged tails through the loader): the same kernels in the same order on the same data."""
    from recbole_cdr_amd.model.cross_domain_recommender.emcdr import EMCDR
    from recbole_cdr_amd.trainer import CrossDomainTrainer
    from recbole_cdr_amd.trainer.trainer import DenseAdam
    from recbole_cdr_amd.utils import InputType, train_mode2state
    ids, ds, s_pairs, t_pairs = _dataset(1)
    cfg = base_config(DEV, latent_factor_model='BPR', source_embedding_size=16, target_embedding_size=16, reg_weight=0.01,
                      mapping_function=mapping, mlp_hidden_size=[24], learning_rate=0.01, train_modes=['SOURCE', 'TARGET', 'OVERLAP'],
                      epoch_num=['2', '2', '2'], source_split=False, eval_step=0, epochs=2, graph_pipeline=graph_pipeline)
    torch.manual_seed(3)
    model = EMCDR(cfg, ds).to(DEV)
    init = _params(model)
    torch.manual_seed(77)
    dl = _loaders(ids, ds, s_pairs, t_pairs, InputType.PAIRWISE, 64, 1, shuffle=True)
    trainer = CrossDomainTrainer(cfg, model)
    log = []
    orig = trainer._train_epoch
    trainer._train_epoch = lambda data, e: (log.append(orig(data, e)) or log[-1])
    trainer.fit(dl)
    assert trainer.graph_stats['captures'] == 3 and trainer.graph_stats['replayed'] > 0
    n_full = 2 * (len(s_pairs) // 64 + len(t_pairs) // 64 + ids.OU // 16)
    n_tail = 2 * ((len(s_pairs) % 64 > 0) + (len(t_pairs) % 64 > 0) + (ids.OU % 16 > 0))
    assert trainer.graph_stats['replayed'] == n_full and trainer.graph_stats['eager'] == n_tail

    # the same sequence without any graph
    torch.manual_seed(3)
    ref = EMCDR(cfg, ds).to(DEV)
    with torch.no_grad():
        for k, v in ref.named_parameters():
            v.copy_(init[k])
    torch.manual_seed(77)
    dl2 = _loaders(ids, ds, s_pairs, t_pairs, InputType.PAIRWISE, 64, 1, shuffle=True)
    opt = DenseAdam(ref.parameters(), lr=0.01)
    ref_log = []

    def step(b):
        opt.zero_grad(set_to_none=True)
        loss = ref.calculate_loss(b)
        loss = loss.reshape(()) if loss.numel() == 1 else loss.sum()
        loss.backward()
        opt.step()
        return loss.detach()
    for phase in ('SOURCE', 'TARGET', 'OVERLAP'):
        dl2.set_mode(train_mode2state[phase])
        ref.set_phase(phase)
        ref.train()
        for _ in range(2):
            it = iter(dl2)
            prod = dl2.device_producer()
            prod.resync()
            tot = torch.zeros((), device=DEV)
            while True:
                if prod.full_ahead():
                    prod.launch()
                    tot += step(prod.fields)
                    prod.advance()
                    continue
                try:
                    b = next(it)
                except StopIteration:
                    break
                tot += step(b)
                prod.resync()
            ref_log.append(float(tot))
    assert len(log) == len(ref_log) == 6
    # (the dense backward adds duplicate rows with fp32 atomics, whose order is not fixed: rounding-level differences are allowed)
    assert_close(torch.tensor(log), torch.tensor(ref_log), rtol=1e-5, what='epoch losses')
    for k, v in model.named_parameters():
        assert_close(v, dict(ref.named_parameters())[k], rtol=1e-4, atol=0.02 * 0.01, what=k)


def test_trainer_epochs_cover_every_interaction_once():
    """An epoch of replays + the loader's ragged tail visits every interaction exactly once (shuffled in place), in the loader's
    k-major layout, and two epochs differ in order and in negatives."""
    from recbole_cdr_amd.utils import InputType, train_mode2state
    ids, ds, s_pairs, t_pairs = _dataset(2)
    torch.manual_seed(5)
    dl = _loaders(ids, ds, s_pairs, t_pairs, InputType.PAIRWISE, 96, 2, shuffle=True)
    dl.set_mode(train_mode2state['TARGET'])
    want = sorted(map(tuple, t_pairs.tolist()))
    seen_epochs = []
    for _ in range(2):
        it = iter(dl)
        prod = dl.device_producer()
        prod.resync()
        rows, negs = [], []
        while True:
            if prod.full_ahead():
                prod.launch(); prod.advance()
                b = prod.fields
            else:
                try:
                    b = next(it)
                except StopIteration:
                    break
                prod.resync()
            n = b['target_user_id'].numel() // 2
            u, i = b['target_user_id'][:n].tolist(), b['target_item_id'][:n].tolist()
            assert b['target_user_id'][n:].tolist() == u and b['target_item_id'][n:].tolist() == i and b.k_major == 2
            rows += list(zip(u, i)); negs += b['neg_target_item_id'].tolist()
        assert sorted(rows) == want
        seen_epochs.append((rows, negs))
    assert seen_epochs[0][0] != seen_epochs[1][0] and seen_epochs[0][1] != seen_epochs[1][1]


def test_trainer_both_mode_source_wraps_under_replays():
    """BOTH state (CMF's only phase): the source loader is shorter than the target loader, wraps without reshuffling
    (dataloader.py:156-161) and the epoch ends with the target loader -- with the wrap batches served by the loader and everything else
    by replays, the run equals the eager trainer on the same loaders."""
    from recbole_cdr_amd.model.cross_domain_recommender.cmf import CMF
    from recbole_cdr_amd.trainer import CrossDomainTrainer
    from recbole_cdr_amd.utils import InputType
    ids, ds, s_pairs, t_pairs = _dataset(3, n_s=300, n_t=900)
    cfg = base_config(DEV, embedding_size=16, alpha=0.4, learning_rate=0.01, train_modes=['BOTH'], epoch_num=['3'], source_split=False,
                      eval_step=0, epochs=3, **{'lambda': 0.01, 'gamma': 0.02})
    outs = []
    for graph in (True, False):
        torch.manual_seed(4)
        model = CMF(cfg, ds).to(DEV)
        torch.manual_seed(9)
        dl = _loaders(ids, ds, s_pairs, t_pairs, InputType.POINTWISE, 128, 1, shuffle=False)
        trainer = CrossDomainTrainer(dict(cfg, graph_step=graph), model)
        log = []
        orig = trainer._train_epoch
        trainer._train_epoch = lambda data, e, o=orig, l=log: (l.append(o(data, e)) or l[-1])
        trainer.fit(dl)
        outs.append((log, _params(model), dict(trainer.graph_stats)))
    (lg, pg, sg), (le, pe, se) = outs
    assert sg['replayed'] > 0 and sg['eager'] > 0 and se['replayed'] == 0
    # shuffle=False: both runs see the same positives; the negatives come from different counter streams (in-graph vs host-counted),
    # so the comparison is statistical on the loss and exact on the schedule
    assert len(lg) == len(le) == 3
    assert sg['replayed'] + sg['eager'] == 3 * ((len(t_pairs) + 63) // 64)
    for a, b in zip(lg, le):
        assert abs(a - b) < 0.1 * abs(b), (lg, le)


def test_conet_trainer_uses_deferred_adam_and_matches_dense():
    """Trainer picks RowAwareAdam for CoNet (exact dense Adam evaluated lazily per row) and replays the step; against the same
    trainer with the literal dense sweep, eagerly: same losses, same parameters after two epochs (bit-identical update rule)."""
    from recbole_cdr_amd.model.cross_domain_recommender.conet import CoNet
    from recbole_cdr_amd.trainer import CrossDomainTrainer
    from recbole_cdr_amd.trainer.trainer import RowAwareAdam, DenseAdam
    from recbole_cdr_amd.utils import InputType
    ids, ds, s_pairs, t_pairs = _dataset(4)
    cfg = base_config(DEV, embedding_size=16, reg_weight=0.01, mlp_hidden_size=[32, 16, 8], learning_rate=0.01, train_modes=['BOTH'],
                      epoch_num=['2'], source_split=False, eval_step=0, epochs=2)
    outs = []
    for fast in (True, False):
        torch.manual_seed(6)
        model = CoNet(cfg, ds).to(DEV)
        rng = {'s': np.random.RandomState(1), 't': np.random.RandomState(2)}
        src_i = np.arange(ids.OI + ids.TOI, ids.total_num_items); tgt_i = np.arange(1, ids.OI + ids.TOI)
        s_smp = lambda u, i, k: torch.from_numpy(rng['s'].choice(src_i, u.numel() * k)).to(u.device)
        t_smp = lambda u, i, k: torch.from_numpy(rng['t'].choice(tgt_i, u.numel() * k)).to(u.device)
        from recbole_cdr_amd.data import CrossDomainDataloader, OverlapDataloader, DomainTrainLoader
        dt = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(DEV)
        dl = CrossDomainDataloader(
            DomainTrainLoader({'source_user_id': dt(s_pairs[:, 0]), 'source_item_id': dt(s_pairs[:, 1])}, 'source_user_id', 'source_item_id',
                              'source_label', 'neg_', 128, 1, InputType.POINTWISE, s_smp),
            DomainTrainLoader({'target_user_id': dt(t_pairs[:, 0]), 'target_item_id': dt(t_pairs[:, 1])}, 'target_user_id', 'target_item_id',
                              'target_label', 'neg_', 128, 1, InputType.POINTWISE, t_smp),
            OverlapDataloader(ids.OU, 8, device=DEV))
        trainer = CrossDomainTrainer(dict(cfg, graph_step=fast, deferred_adam=fast), model)
        assert isinstance(trainer.optimizer, RowAwareAdam if fast else DenseAdam)
        log = []
        orig = trainer._train_epoch
        trainer._train_epoch = lambda data, e, o=orig, l=log: (l.append(o(data, e)) or l[-1])
        trainer.fit(dl)
        outs.append((log, {k: v.detach().clone() for k, v in model.state_dict().items()}, dict(trainer.graph_stats)))
    (lf, pf, sf), (ld, pd, sd) = outs
    assert sf['replayed'] > 0 and sd['replayed'] == 0
    assert_close(torch.tensor(lf), torch.tensor(ld), rtol=1e-5, what='epoch losses')
    for k in pf:
        assert_close(pf[k], pd[k], rtol=1e-4, atol=0.01 * 1e-2, what=k)


# ---- cdr_conet_fullsort (conet.py:222-242) ---------------------------------------------------------------------
@pytest.mark.parametrize('D,layers,U,N', [(16, [64, 32, 16, 8], 5, 77), (128, [64, 32, 16, 8], 64, 2000), (8, [12, 8, 4], 3, 41),
                                          (32, [48, 24], 7, 100), (16, [20, 32, 32, 32], 9, 65), (64, [64, 8], 33, 1)])
def test_conet_fullsort_kernel_vs_fp64_tower(D, layers, U, N):
    """All users x all items through the target tower in one launch against an fp64 evaluation of the reference's formula
    (cat([u, i]) -> Linear + ReLU per layer -> Linear + Sigmoid), 1e-5 relative; shapes: the default tower, C3's D = 128, the golden
    'a' tower, two- and four-layer towers, widths that are not multiples of 8, a one-item catalogue."""
    from recbole_cdr_amd import functional as F_
    g = torch.Generator().manual_seed(D + N)
    dims = [2 * D] + layers
    Ws = [torch.randn(b, a, generator=g) * (2.0 / (a + b)) ** 0.5 for a, b in zip(dims[:-1], dims[1:])]
    bs = [torch.randn(b, generator=g) * 0.1 for b in dims[1:]]
    wo, bo = torch.randn(1, dims[-1], generator=g) * 0.5, torch.randn(1, generator=g) * 0.1
    users, items = torch.randn(U, D, generator=g) * 0.3, torch.randn(N, D, generator=g) * 0.3
    x = torch.cat([users.double().unsqueeze(1).expand(U, N, D), items.double().unsqueeze(0).expand(U, N, D)], dim=2)
    for W, b in zip(Ws, bs):
        x = torch.relu(x @ W.double().t() + b.double())
    want = torch.sigmoid(x @ wo.double().t() + bo.double()).squeeze(-1)
    dv = lambda t: t.to(DEV)
    P = F_.gemm(dv(items), dv(Ws[0])[:, D:], trans_b=True)
    Q = F_.gemm(dv(users), dv(Ws[0])[:, :D], trans_b=True, bias=dv(bs[0]))
    assert F_.conet_fullsort_supported(layers[0], layers[1:])
    got = F_.conet_fullsort(P, Q, [dv(w) for w in Ws[1:]], [dv(b) for b in bs[1:]], dv(wo), dv(bo))
    torch.cuda.synchronize()
    assert tuple(got.shape) == (U, N)
    assert_close(got, want.float(), rtol=1e-5, what='conet fullsort')


@pytest.mark.parametrize('D,layers,U,N', [(16, [64, 32, 16, 8], 1, 77), (128, [64, 32, 16, 8], 1, 5000), (8, [12, 8, 4], 3, 41), (32, [48, 24], 8, 100),
                                          (20, [20, 32, 32, 32], 5, 65), (64, [64, 8], 40, 33)])
def test_conet_fullsort_users_entry_vs_fp64_tower(D, layers, U, N):
    """cdr_conet_fullsort_users (the few-users call of recbole's evaluation loop as ONE launch: the user half of the first layer formed
    inside it from the table row) against the fp64 tower, 1e-5 relative; a padded table (row stride > D), repeated ids, and more users
    than one LDS chunk."""
    from recbole_cdr_amd import functional as F_
    g = torch.Generator().manual_seed(D + N + U)
    dims = [2 * D] + layers
    Ws = [torch.randn(b, a, generator=g) * (2.0 / (a + b)) ** 0.5 for a, b in zip(dims[:-1], dims[1:])]
    bs = [torch.randn(b, generator=g) * 0.1 for b in dims[1:]]
    wo, bo = torch.randn(1, dims[-1], generator=g) * 0.5, torch.randn(1, generator=g) * 0.1
    table = torch.randn(50, D + 4, generator=g) * 0.3
    uid = torch.randint(0, 50, (U,), generator=g)
    if U > 2:
        uid[1] = uid[0]
    users, items = table[uid, :D], torch.randn(N, D, generator=g) * 0.3
    x = torch.cat([users.double().unsqueeze(1).expand(U, N, D), items.double().unsqueeze(0).expand(U, N, D)], dim=2)
    for W, b in zip(Ws, bs):
        x = torch.relu(x @ W.double().t() + b.double())
    want = torch.sigmoid(x @ wo.double().t() + bo.double()).squeeze(-1)
    dv = lambda t: t.to(DEV)
    W1 = dv(Ws[0])
    P = F_.gemm(dv(items), W1[:, D:], trans_b=True)
    call = F_.ConetFullsortFewUsers(P, dv(table)[:, :D], W1, dv(bs[0]), D, [dv(w) for w in Ws[1:]], [dv(b) for b in bs[1:]], dv(wo), dv(bo))
    got = call(dv(uid))
    torch.cuda.synchronize()
    assert tuple(got.shape) == (U, N)
    assert_close(got, want.float(), rtol=1e-5, what='conet fullsort users')


def test_conet_full_sort_predict_few_users_in_eval_mode_is_the_one_launch_call_and_follows_training():
    """CoNet.full_sort_predict in evaluation mode: the first call builds P and the packed few-users call, later calls with <= 8 users are
    that one launch -- equal to the general path to 1e-5; a training step in between drops the cache (the scores follow the new weights)."""
    from oracle.common import IdSpace
    from recbole_cdr_amd.model.cross_domain_recommender.conet import CoNet
    ids = IdSpace(OU=50, TOU=30, SOU=20, OI=1, TOI=2999, SOI=500)
    cfg = base_config(DEV, embedding_size=32, reg_weight=0.01, mlp_hidden_size=[64, 32, 16, 8])
    torch.manual_seed(2)
    model = CoNet(cfg, FakeDataset(ids)).to(DEV)
    model.eval()
    one = {'target_user_id': torch.tensor([7], device=DEV)}
    many = {'target_user_id': torch.arange(1, 41, device=DEV)}
    with torch.no_grad():
        # outside a freeze_for_eval() bracket nothing is cached: an in-place parameter change under eval() is seen by the next call, as in
        # the reference -- through the tensor, through .data and through a raw-pointer write alike
        s0 = model.full_sort_predict(one)
        assert '_eval_few' not in model.__dict__ and '_eval_P' not in model.__dict__
        model.target_item_embedding.weight.data[5].mul_(3.0)
        model.target_crossunit_linear[0].weight.data.mul_(1.25)
        s1 = model.full_sort_predict(one)
        assert float((s1 - s0).abs().max()) > 0
        import copy
        import pickle
        model.freeze_for_eval()
        ref_many = model.full_sort_predict(many)                      # general path; builds the caches
        assert '_eval_few' in model.__dict__ and '_eval_P' in model.__dict__
        twin = copy.deepcopy(model)                                   # a best-model copy after validation: the caches (ctypes pointers) stay behind
        assert '_eval_few' not in twin.__dict__ and '_eval_P' not in twin.__dict__ and not twin._eval_caching()
        assert_close(twin.full_sort_predict(one), ref_many[6:7], rtol=1e-5, what='deep copy scores')
        pickle.loads(pickle.dumps(model))
        got = model.full_sort_predict(one)                            # one launch
        assert_close(got, ref_many[6:7], rtol=1e-5, what='few users vs general path')
        five = {'target_user_id': torch.tensor([3, 9, 9, 40, 1], device=DEV)}
        assert_close(model.full_sort_predict(five), ref_many[[2, 8, 8, 39, 0]], rtol=1e-5, what='five users')
        model.train()
        assert '_eval_few' not in model.__dict__ and '_eval_P' not in model.__dict__
        for p in model.target_crossunit_linear[0].parameters():
            p.mul_(0.5)
        model.target_user_embedding.weight[7].mul_(2.0)
        model.eval()
        assert not model._eval_caching()                              # train() closed the bracket
        model.freeze_for_eval()
        a = model.full_sort_predict(one)                              # general path again (rebuilds), then the one-launch call
        b = model.full_sort_predict(one)
        assert_close(b, a, rtol=1e-5, what='after training: few users vs general path')
        assert float((a - got).abs().max()) > 1e-4                    # (the weights did change)


def test_conet_full_sort_predict_fused_equals_layerwise_path():
    """CoNet.full_sort_predict on the one-launch kernel against the same model's per-user contraction path (the round-3 product
    path, still what towers outside the kernel's range take): C3-shaped tower, 40 users x 3,000 items."""
    from oracle.common import IdSpace
    from recbole_cdr_amd.model.cross_domain_recommender.conet import CoNet
    ids = IdSpace(OU=50, TOU=30, SOU=20, OI=1, TOI=2999, SOI=500)
    cfg = base_config(DEV, embedding_size=32, reg_weight=0.01, mlp_hidden_size=[64, 32, 16, 8])
    torch.manual_seed(1)
    model = CoNet(cfg, FakeDataset(ids)).to(DEV)
    with torch.no_grad():
        for n, p in model.named_parameters():
            if n.endswith('bias'):
                p.normal_(0, 0.1)
    ev = {'target_user_id': torch.arange(1, 41, device=DEV)}
    fused = model.full_sort_predict(ev)
    model.__dict__['fullsort_fused'] = False
    loop = model.full_sort_predict(ev)
    assert fused.shape == loop.shape == (40, ids.OI + ids.TOI)
    assert_close(fused, loop, rtol=1e-5, what='fused vs layerwise')


# ---- full-size parity of the instantiations the bench times (emcdr.py:110-154) ---------------------------------
def _rows_close(got, want, what, rtol=1e-5, abs_floor=0.0):
    """assert_close's matrix rule on the device (the operands here are up to 1 M x 128): |got - want| <= rtol |want| + 1e-5 x
    max(row max, 1e-3 x tensor max)."""
    big = float(want.abs().max())
    floor = 1e-5 * torch.maximum(want.abs().amax(1, keepdim=True), torch.tensor(1e-3 * big, device=want.device)) + 1e-12
    bad = (got - want).abs() > rtol * want.abs() + floor + abs_floor
    n_bad = int(bad.sum())
    assert n_bad == 0, f'{what}: {n_bad} / {bad.numel()} elements beyond {rtol:g} (row-scaled floor)'


@pytest.mark.parametrize('kind', ['adam', 'kmajor4', 'zipf'])
def test_full_c5_size_adam_instantiations(kind):
    """BASELINE C5 table sizes (50,000,001 x 128 users, 20,000,001 x 128 items, 1,048,576 triples) with the ADAM instantiations the
    bench times -- cdr_bpr_step_fused (bpr_fwd_apply_kernel<32,1,1>), cdr_bpr_step_fused_kmajor at k = 4, and the Zipf(1.05) batch
    with its long duplicate segments -- through properties that need no oracle run at that size:
      * lr = 0: every weight bit-identical; the moments become non-zero on exactly the batch's rows and nowhere else;
      * first-step moments against an INDEPENDENT run: m = (1 - b1) g and v = (1 - b2) g^2 where g is recovered from the weight
        change of the plain SGD step (FusedBPRStep(opt='sgd'), the instantiation round 3 already covers) on the same triples;
      * a real first Adam step equals -lr g / (|g| + eps) element by element (g from that SGD run), and twice from the same start is
        bit-identical in weights, both moments and the loss."""
    from recbole_cdr_amd.fused import FusedBPRStep, KMajorBPRStep, RowwiseState, OPT_ADAM
    free_b, _ = torch.cuda.mem_get_info()
    if free_b < 170e9:
        pytest.skip('needs ~150 GB of free HBM')
    nu, ni, D, B, TOI = 50_000_001, 20_000_001, 128, 1 << 20, 10_000_000
    b1, b2 = 0.9, 0.999
    g = torch.Generator(device=DEV); g.manual_seed(11)
    U = torch.empty(nu, D, device=DEV).normal_(0, 0.05, generator=g)
    I = torch.empty(ni, D, device=DEV).normal_(0, 0.05, generator=g)
    k = 4 if kind == 'kmajor4' else 1
    S = B // k
    u = torch.randint(1, nu, (S,), device=DEV, generator=g)
    if kind == 'zipf':
        r = torch.rand(S, device=DEV, generator=g, dtype=torch.float64)
        p = 1 + ((float(TOI) ** (1 - 1.05) - 1) * r + 1).pow(1 / (1 - 1.05)).long().clamp_(1, TOI) - 1
        assert int(torch.bincount(p).max()) > 10000                     # the hottest item occurs tens of thousands of times
    else:
        p = torch.randint(1, 1 + TOI, (S,), device=DEV, generator=g)
    n = torch.randint(1, 1 + TOI, (B,), device=DEV, generator=g)
    uf, pf = u.repeat(k), p.repeat(k)                                    # the triples, one per row (k-major)
    ust, ist = RowwiseState(U, OPT_ADAM), RowwiseState(I, OPT_ADAM)

    def make(lr):
        if kind == 'kmajor4':
            st = KMajorBPRStep(U, I, S, k=4, opt='adam', lr=lr, reg_weight=0.0, user_state=ust, item_state=ist)
            assert st.fuse_singles
            return lambda: st.step(u, p, n)[0].clone()
        st = FusedBPRStep(U, I, B, opt='adam', lr=lr, reg_weight=0.0, user_state=ust, item_state=ist)
        assert st.fuse_singles
        return lambda: st.step(uf, pf, n)[0].clone()
    tu, ti = torch.unique(uf), torch.unique(torch.cat([pf, n]))
    Wu0, Wi0 = U[tu].clone(), I[ti].clone()
    # ---- lr = 0 -----------------------------------------------------------------------------------------------------------------
    loss0 = make(0.0)()
    assert torch.equal(U[tu], Wu0) and torch.equal(I[ti], Wi0)
    for st, touched, name in ((ust, tu, 'user'), (ist, ti, 'item')):
        for mom in (st.exp_avg, st.exp_avg_sq):
            nz = torch.nonzero(mom.abs().amax(1) > 0).flatten()
            assert bool(torch.isin(nz, touched).all()), f'{name} moments written outside the batch'
            assert touched.numel() - 3 <= nz.numel() <= touched.numel(), (name, nz.numel(), touched.numel())   # (p == n cancels a row: ~0.1 expected)
            del nz
    mu, vu, mi, vi = ust.exp_avg[tu].clone(), ust.exp_avg_sq[tu].clone(), ist.exp_avg[ti].clone(), ist.exp_avg_sq[ti].clone()
    # ---- the gradient from an independent instantiation: plain SGD step on the same triples --------------------------------------
    lr_s = float(B)
    FusedBPRStep(U, I, B, opt='sgd', lr=lr_s, reg_weight=0.0).step(uf, pf, n)
    gu, gi = (Wu0 - U[tu]) / lr_s, (Wi0 - I[ti]) / lr_s
    U[tu] = Wu0; I[ti] = Wi0
    _rows_close(mu, (1 - b1) * gu, 'user exp_avg vs (1 - b1) g')
    _rows_close(mi, (1 - b1) * gi, 'item exp_avg vs (1 - b1) g')
    _rows_close(vu, (1 - b2) * gu * gu, 'user exp_avg_sq vs (1 - b2) g^2', rtol=3e-5)
    _rows_close(vi, (1 - b2) * gi * gi, 'item exp_avg_sq vs (1 - b2) g^2', rtol=3e-5)
    del mu, vu, mi, vi
    # ---- a real first Adam step, twice from the same start ------------------------------------------------------------------------
    lr = 1e-3
    runs = []
    for _ in range(2):
        for st, touched in ((ust, tu), (ist, ti)):
            st.exp_avg[touched] = 0; st.exp_avg_sq[touched] = 0
            st.step = 0
        U[tu] = Wu0; I[ti] = Wi0
        loss = make(lr)()
        runs.append((loss, U[tu].clone(), I[ti].clone(), ust.exp_avg[tu].clone(), ist.exp_avg_sq[ti].clone()))
    assert all(torch.equal(a, b) for a, b in zip(runs[0], runs[1])), 'the Adam step is not reproducible'
    assert torch.equal(runs[0][0], loss0)                                  # same forward
    # first Adam step from zero moments: m_hat = g, v_hat = g^2, so dw = -lr g / (|g| + eps) -- with g from the independent SGD run
    # (|g| ~ 2e-8 here, the same order as eps = 1e-8: the eps term is exercised, |dw| ~ 0.7 lr)
    # (both sides are differences of fp32 weights ~0.05: one ulp of those, 3.7e-9, is the absolute resolution -- 2e-8 = 2e-5 of the step)
    _rows_close(runs[0][1] - Wu0, -lr * gu / (gu.abs() + 1e-8), 'user first Adam step', rtol=1e-4, abs_floor=2e-8)
    _rows_close(runs[0][2] - Wi0, -lr * gi / (gi.abs() + 1e-8), 'item first Adam step', rtol=1e-4, abs_floor=2e-8)


def test_parallel_domains_on_two_streams_equals_sequential_phases():
    """config['parallel_domains'] on one GPU (rowwise mode): the SOURCE and TARGET epochs enqueued side by side on two HIP streams leave
    exactly the tables, moments and epoch losses of the two phases run one after the other (disjoint state; same batches: the device
    loaders draw from per-loader generators and per-sampler counters)."""
    from recbole_cdr_amd.model.cross_domain_recommender.emcdr import EMCDR
    from recbole_cdr_amd.trainer import CrossDomainTrainer
    from recbole_cdr_amd.utils import InputType
    ids, ds, s_pairs, t_pairs = _dataset(7, n_s=900, n_t=700)
    base = base_config(DEV, latent_factor_model='BPR', source_embedding_size=16, target_embedding_size=16, reg_weight=0.01,
                       mapping_function='linear', mlp_hidden_size=[24], learning_rate=0.01, train_modes=['SOURCE', 'TARGET', 'OVERLAP'],
                       epoch_num=['3', '2', '1'], source_split=False, eval_step=0, epochs=3, optimizer_mode='rowwise')
    outs = []
    for par in (True, False):
        torch.manual_seed(3)
        model = EMCDR(base, ds).to(DEV)
        dl = _loaders(ids, ds, s_pairs, t_pairs, InputType.PAIRWISE, 128, 1, shuffle=True)
        trainer = CrossDomainTrainer(dict(base, parallel_domains=par), model)
        log = []
        orig = trainer._train_epoch
        trainer._train_epoch = lambda data, e, o=orig, l=log: (l.append(o(data, e)) or l[-1])
        if par:
            orig2 = trainer._fit_domains_on_two_streams
            trainer._fit_domains_on_two_streams = lambda d, p, o=orig2, l=log: l.append(o(d, p))
        trainer.fit(dl)
        torch.cuda.synchronize()
        outs.append((log, {k: v.detach().clone() for k, v in model.named_parameters()}, model.fused_optimizer_state()))
    (lp, pp, sp), (ls, ps, ss) = outs
    assert isinstance(lp[0], dict) and lp[0]['SOURCE'] == ls[:3] and lp[0]['TARGET'] == ls[3:5] and lp[1:] == ls[5:]
    for k in pp:
        assert torch.equal(pp[k], ps[k]), k
    for name in sp['tables']:
        assert sp['tables'][name]['step'] == ss['tables'][name]['step']
        assert torch.equal(sp['tables'][name]['exp_avg'], ss['tables'][name]['exp_avg']), name


def test_conet_pipelined_unrolled_graph_is_bit_identical_to_the_plain_order():
    """CoNet through CrossDomainTrainer on a device loader: the 8-step graph software-pipelined over two streams (row update of step i,
    production + id sort + postponed-update replay of batch i+1 beside the weight gradients and the dense Adam of step i) against the same
    graph in the plain launch order: the same kernels on the same operands -- tables, moments, tower weights and epoch losses bit-equal."""
    from recbole_cdr_amd.model.cross_domain_recommender.conet import CoNet
    from recbole_cdr_amd.trainer import CrossDomainTrainer
    from recbole_cdr_amd.utils import InputType
    ids, ds, s_pairs, t_pairs = _dataset(9, n_s=2400, n_t=2400)
    cfg = base_config(DEV, embedding_size=16, reg_weight=0.01, mlp_hidden_size=[32, 16, 8], learning_rate=0.01, train_modes=['BOTH'],
                      epoch_num=['3'], source_split=False, eval_step=0, epochs=3)
    outs = []
    for pipe in (True, False, 'two_ahead'):
        torch.manual_seed(6)
        model = CoNet(cfg, ds).to(DEV)
        dl = _loaders(ids, ds, s_pairs, t_pairs, InputType.POINTWISE, 128, 1, shuffle=True)
        trainer = CrossDomainTrainer(dict(cfg, graph_pipeline=pipe), model)
        log = []
        orig = trainer._train_epoch
        trainer._train_epoch = lambda data, e, o=orig, l=log: (l.append(o(data, e)) or l[-1])
        trainer.fit(dl)
        gs = [g for g in trainer._graphs.values() if g][0]
        assert bool(gs._can_pipeline()) == bool(pipe) and gs.unroll == (16 if pipe else 8) and trainer.graph_stats['replayed'] >= 3 * 16
        assert (gs._statics is not None) == (pipe == 'two_ahead')          # (the two-batches-deep order really ran on its second batch slot)
        outs.append((log, {k: v.detach().clone() for k, v in model.state_dict().items()}, trainer.optimizer.state_dict()))
    (ls, ps, os_) = outs[1]                                                  # the plain order
    for (lp, pp, op) in (outs[0], outs[2]):
        assert lp == ls, (lp, ls)
        for k in pp:
            assert torch.equal(pp[k], ps[k]), k
        assert 'deferred_rows' not in op and op['state'].keys() == os_['state'].keys()
        for i_ in op['state']:
            for f_ in ('exp_avg', 'exp_avg_sq', 'step'):
                assert torch.equal(op['state'][i_][f_], os_['state'][i_][f_]), (i_, f_)


def test_rowwise_trainer_on_replays_equals_the_eager_rowwise_loop():
    """optimizer_mode='rowwise' on device loaders: the per-triple BPR step (update counts on the device: cdr_bpr_step_fused_dev) and the
    distinct-id OVERLAP step replayed as hipGraphs -- {producer -> fused step -> loss total}, four steps per launch after two eager steps
    -- against the same trainer with graph_step=False: same producer launches, same kernels: tables, moments, mapping, update counts and
    epoch losses bit-equal."""
    from oracle.common import IdSpace
    from recbole_cdr_amd.model.cross_domain_recommender.emcdr import EMCDR
    from recbole_cdr_amd.trainer import CrossDomainTrainer
    from recbole_cdr_amd.utils import InputType
    ids = IdSpace(OU=3000, TOU=500, SOU=400, OI=1, TOI=2500, SOI=2200)
    rng = np.random.RandomState(3)
    src_u = np.r_[1:ids.OU, ids.OU + ids.TOU:ids.total_num_users]
    s_pairs = np.unique(np.stack([rng.choice(src_u, 120000), rng.randint(ids.OI + ids.TOI, ids.total_num_items, 120000)], 1), axis=0)
    t_pairs = np.unique(np.stack([rng.randint(1, ids.OU + ids.TOU, 120000), rng.randint(1, ids.OI + ids.TOI, 120000)], 1), axis=0)
    rng.shuffle(s_pairs); rng.shuffle(t_pairs)
    ds = FakeDataset(ids, s_pairs, t_pairs)
    cfg = base_config(DEV, latent_factor_model='BPR', source_embedding_size=32, target_embedding_size=32, reg_weight=0.01,
                      mapping_function='linear', mlp_hidden_size=[24], learning_rate=0.01, train_modes=['SOURCE', 'TARGET', 'OVERLAP'],
                      epoch_num=['2', '1', '2'], source_split=False, eval_step=0, epochs=2, optimizer_mode='rowwise')
    outs = []
    for graph in (True, False):
        torch.manual_seed(3)
        model = EMCDR(cfg, ds).to(DEV)
        dl = _loaders(ids, ds, s_pairs, t_pairs, InputType.PAIRWISE, 8192, 1, shuffle=True, ob=256)
        trainer = CrossDomainTrainer(dict(cfg, graph_step=graph), model)
        log = []
        orig = trainer._train_epoch
        trainer._train_epoch = lambda data, e, o=orig, l=log: (l.append(o(data, e)) or l[-1])
        trainer.fit(dl)
        torch.cuda.synchronize()
        outs.append((log, {k: v.detach().clone() for k, v in model.named_parameters()}, model.fused_optimizer_state(), dict(trainer.graph_stats)))
    (lg, pg, sg, stg), (le, pe, se, ste) = outs
    assert stg['captures'] == 3 and stg['replayed'] > 20 and ste['replayed'] == 0
    assert lg == le, (lg, le)
    for k in pg:
        assert torch.equal(pg[k], pe[k]), k
    for name in sg['tables']:
        assert sg['tables'][name]['step'] == se['tables'][name]['step'], name
        assert torch.equal(sg['tables'][name]['exp_avg'], se['tables'][name]['exp_avg']), name


def _bench_c3_state(extra=(), env=None):
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    import tempfile
    with tempfile.TemporaryDirectory() as tmp:
        detail = os.path.join(tmp, 'detail.json')
        cmd = [sys.executable, os.path.join(root, 'bench.py'), '--workload', 'c3', '--no-cpu-baseline', '--no-fullsort', '--steps', '40', '--warmup', '4',
               '--detail-file', detail] + list(extra)
        p = subprocess.run(cmd, cwd=root, env=dict(os.environ, **(env or {})), capture_output=True, text=True, timeout=600)
        assert p.returncode == 0, p.stderr[-3000:]
        line = json.loads([l for l in p.stdout.splitlines() if l.strip()][-1])          # the bounded contract line ...
        d = json.load(open(detail))                                                       # ... and every leg, in the file it names
        assert line['detail_file'] == detail and line['value'] == d['value']
    return d['state_checksum'], d['final_loss'], d['config']['trainer_steps']


@pytest.mark.gpu
def test_c3_trained_state_is_the_dense_sweeps_in_every_launch_order_and_process():
    """BASELINE C3 (CoNet, 156 k users x 134 k items, D = 128, 8,190 rows per step) through ``CrossDomainTrainer.fit``, 40 + 40 steps, each
    variant in a process of its own: the literal dense Adam sweep; the deferred Adam in the plain launch order; pipelined one batch ahead
    (the default) and two batches ahead.  The exact fp64 sums of every trained parameter agree across all of them, and a repeat of the
    default run prints the same digits.  (This is the check that found the wave race in ``lz_prepare1_kernel``: the deferred runs differed
    from the dense sweep -- and from each other -- in the seventh digit.)"""
    dense, loss_d, _ = _bench_c3_state(['--dense-adam'])
    runs = {'plain': _bench_c3_state(env={'CDR_GRAPH_PIPELINE': '0'}),
            'one_ahead': _bench_c3_state(),
            'one_ahead again': _bench_c3_state(),
            'two_ahead': _bench_c3_state(env={'CDR_GRAPH_PIPELINE': 'two_ahead'})}
    for name, (state, loss, steps) in runs.items():
        assert steps['optimizer'] == 'RowAwareAdam' and steps['replayed'] == 40, (name, steps)
        assert state == dense, (name, {k: (state[k], dense[k]) for k in dense if state[k] != dense[k]})
        assert loss == loss_d, (name, loss, loss_d)


@pytest.mark.gpu
def test_headline_step_trains_the_same_tables_in_every_process_and_stream_layout():
    """``bench.py --headline-only`` (EMCDR-BPR fused row-wise step, Zipf-free uniform batches with duplicate rows) three times in processes of
    their own -- two domain streams, again, and one stream: the exact fp64 sums of the four trained tables and the last loss agree."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = []
    import tempfile
    for extra in ([], [], ['--single-stream']):
        with tempfile.TemporaryDirectory() as tmp:
            detail = os.path.join(tmp, 'detail.json')
            cmd = [sys.executable, os.path.join(root, 'bench.py'), '--headline-only', '--steps', '6', '--warmup', '2', '--users', '2000001',
                   '--items-per-domain', '300000', '--batch', '262144', '--detail-file', detail] + extra
            p = subprocess.run(cmd, cwd=root, capture_output=True, text=True, timeout=600)
            assert p.returncode == 0, p.stderr[-3000:]
            d = json.load(open(detail))
        outs.append((d['state_checksum'], d['final_loss']))
    assert outs[0] == outs[1] == outs[2], outs
    assert len(outs[0][0]) == 4 and all(float(v) == float(v) for v in outs[0][0].values())


def test_bitgcf_evaluation_after_replayed_epochs_sees_the_trained_tables():
    """BiTGCF caches its propagated target embeddings for evaluation and the reference clears that cache at the top of
    every calculate_loss (bitgcf.py:146-148) -- Python a hipGraph replay never runs.  With every batch of an epoch a full-shape replay,
    the evaluation after the NEXT epoch must still see the trained tables (Trainer calls model.on_train_steps() after replays)."""
    from recbole_cdr_amd.model.cross_domain_recommender.bitgcf import BiTGCF
    from recbole_cdr_amd.trainer import CrossDomainTrainer
    from recbole_cdr_amd.utils import InputType
    ids, ds, s_pairs, t_pairs = _dataset(7, n_s=900, n_t=900)
    s_pairs, t_pairs = s_pairs[:384], t_pairs[:384]                      # 6 full batches of 64 positives (+64 negatives) per epoch: no ragged tail
    ds = FakeDataset(ids, s_pairs, t_pairs)
    cfg = base_config(DEV, embedding_size=16, n_layers=2, reg_weight=1e-3, lambda_source=0.8, lambda_target=0.7, drop_rate=0.0,
                      connect_way='concat', learning_rate=0.05, train_modes=['BOTH'], epoch_num=['1'], source_split=False, eval_step=0, epochs=1)
    ev = {'target_user_id': torch.arange(1, 9, device=DEV)}
    scores = {}
    for graph in (True, False):
        torch.manual_seed(4)
        model = BiTGCF(cfg, ds).to(DEV)
        dl = _loaders(ids, ds, s_pairs, t_pairs, InputType.POINTWISE, 128, 1, shuffle=False)
        trainer = CrossDomainTrainer(dict(cfg, graph_step=graph), model)
        trainer.fit(dl)                                                   # epoch 0 (graph: warm-up + capture + replays)
        model.eval()
        first = model.full_sort_predict(ev).clone()                       # fills the cache
        before = dict(trainer.graph_stats)
        trainer.fit(dl)                                                   # epoch 1: with graph_step every batch is a replay
        after = dict(trainer.graph_stats)
        model.eval()
        got = model.full_sort_predict(ev).clone()
        model.init_restore_e()
        fresh = model.full_sort_predict(ev).clone()
        assert torch.equal(got, fresh), 'evaluation reused embeddings propagated before the last epoch'
        assert not torch.allclose(got, first), 'an epoch at lr 0.05 must move the scores'
        if graph:
            assert after['replayed'] - before['replayed'] == 6 and after['eager'] == before['eager'], (before, after)
        scores[graph] = got


def test_row_aware_adam_checkpoint_is_the_torch_adam_layout_both_ways():
    """RowAwareAdam.state_dict() keeps the embedding tables' moments under the standard per-parameter 'state'
    entries.  A CoNet run checkpointed under the deferred form resumes under DenseAdam (and the reverse) and a further epoch lands on
    the parameters of the uninterrupted run; torch.optim.Adam loads the same dict too."""
    from recbole_cdr_amd.model.cross_domain_recommender.conet import CoNet
    from recbole_cdr_amd.trainer import CrossDomainTrainer
    from recbole_cdr_amd.trainer.trainer import RowAwareAdam, DenseAdam
    from recbole_cdr_amd.utils import InputType
    ids, ds, s_pairs, t_pairs = _dataset(11)
    cfg = base_config(DEV, embedding_size=16, reg_weight=0.01, mlp_hidden_size=[32, 16, 8], learning_rate=0.01, train_modes=['BOTH'],
                      epoch_num=['1'], source_split=False, eval_step=0, epochs=1, graph_step=False)

    def run(first_deferred, second_deferred, resume):
        torch.manual_seed(6)
        model = CoNet(cfg, ds).to(DEV)
        dl = _loaders(ids, ds, s_pairs, t_pairs, InputType.POINTWISE, 128, 1, shuffle=False, seed=3)
        tr = CrossDomainTrainer(dict(cfg, deferred_adam=first_deferred), model)
        assert isinstance(tr.optimizer, RowAwareAdam if first_deferred else DenseAdam)
        tr.fit(dl)
        if resume:
            sd_opt, sd_model = tr.optimizer.state_dict(), {k: v.clone() for k, v in model.state_dict().items()}
            assert 'deferred_rows' not in sd_opt
            n_params = sum(len(g['params']) for g in sd_opt['param_groups'])
            assert len(sd_opt['state']) == n_params, 'every parameter (tables included) carries its moments'
            torch.optim.Adam(CoNet(cfg, ds).to(DEV).parameters()).load_state_dict(sd_opt)       # the reference's optimizer accepts it
            torch.manual_seed(6)
            model = CoNet(cfg, ds).to(DEV)
            model.load_state_dict(sd_model)
            tr2 = CrossDomainTrainer(dict(cfg, deferred_adam=second_deferred), model)
            assert isinstance(tr2.optimizer, RowAwareAdam if second_deferred else DenseAdam)
            tr2.optimizer.load_state_dict(sd_opt)
            tr = tr2
        tr.fit(dl)                                              # (shuffle=False + the sampler's counter stream continue: same batches either way)
        if hasattr(tr.optimizer, 'row_opt'):
            tr.optimizer.row_opt.flush()
        return {k: v.detach().clone() for k, v in model.state_dict().items()}

    want = run(True, True, resume=False)
    for a, b in ((True, False), (False, True), (True, True)):
        got = run(a, b, resume=True)
        for k in want:
            assert_close(got[k], want[k], rtol=1e-5, atol=1e-7, what=f'{a}->{b}:{k}')
