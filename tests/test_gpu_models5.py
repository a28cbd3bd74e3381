"""GPU parity (-m gpu) of the five models SURVEY.md 8f-4 lists after the BASELINE configs -- CLFM, DTCDR (NeuMF), DeepAPF, NATR,
DCDCSR -- against (1) golden vectors produced by the reference's own code and (2) the oracle at the models' default sizes.
Tolerance: 1e-5 relative (north_star) for fp32 losses / gradients / scores; history matrices bit-exact."""
import numpy as np
import pytest
import torch

from golden_util import Golden, cases
from helpers import DEV, FakeDataset, base_config, load_params, to_dev, assert_close
from test_gpu_parity import _check_grads

pytestmark = pytest.mark.gpu


def _ds(g, pairs=False):
    ids = g.idspace()
    ds = FakeDataset(ids, g['aux/s_pairs'], g['aux/t_pairs']) if pairs else FakeDataset(ids)
    ds.device = DEV
    return ids, ds


@pytest.mark.parametrize('name', cases('clfm_'))
def test_clfm_golden(name):
    from recbole_cdr_amd.model.cross_domain_recommender.clfm import CLFM
    g = Golden(name)
    ids, ds = _ds(g)
    cfg = base_config(DEV, user_embedding_size=int(g.meta('user_embedding_size')),
                      source_item_embedding_size=int(g.meta('item_embedding_size')),
                      target_item_embedding_size=int(g.meta('item_embedding_size')),
                      share_embedding_size=int(g.meta('share_embedding_size')), alpha=float(g.meta('alpha')),
                      reg_weight=float(g.meta('reg_weight')))
    model = CLFM(cfg, ds).to(DEV)
    load_params(model, g.group('param'))
    inter = to_dev(g.group('in'), DEV)
    loss = model.calculate_loss(inter)
    assert_close(loss, g['loss/BOTH'], what=f'{name}:loss')
    loss.sum().backward()
    _check_grads(model, g, 'BOTH')
    ev = to_dev(g.group('evalin'), DEV)
    assert_close(model.predict(ev), g['predict/BOTH'], what='predict')
    assert_close(model.full_sort_predict(ev), g['fullsort/BOTH'], what='fullsort')


@pytest.mark.parametrize('name', cases('dtcdr_'))
def test_dtcdr_golden(name):
    from recbole_cdr_amd.model.cross_domain_recommender.dtcdr import DTCDR
    g = Golden(name)
    ids, ds = _ds(g)
    cfg = base_config(DEV, embedding_size=int(g.meta('D')), mlp_hidden_size=[int(x) for x in g.meta('mlp_hidden_size')],
                      dropout_prob=0.0, base_model='NeuMF', alpha=float(g.meta('alpha')))
    model = DTCDR(cfg, ds).to(DEV)
    load_params(model, g.group('param'))
    model.train()
    loss = model.calculate_loss(to_dev(g.group('in'), DEV))
    assert_close(loss, g['loss/BOTH'], what=f'{name}:loss')
    loss.backward()
    # the predict layers' bias gradient is ONE number: the signed sum over the batch of (p - y) / B terms of magnitude ~3e-2 that
    # cancel to ~1e-4 -- it is held to 1e-5 of its TERMS; everything else goes through the element-wise check
    want = g.group('grad/BOTH', as_torch=False)
    for n_ in [k for k in want if k.endswith('predict_layer.bias')]:
        got = dict(model.named_parameters())[n_].grad
        assert_close(got, want[n_], atol=1e-5 * 3e-2, what=n_)
        dict(model.named_parameters())[n_].grad = torch.from_numpy(want[n_]).to(DEV)
    _check_grads(model, g, 'BOTH')                 # includes rows with exact ties of torch.maximum (gradient split 1/2 : 1/2)
    model.eval()
    assert_close(model.predict(to_dev(g.group('evalin'), DEV)), g['predict/BOTH'], what='predict')


def test_dtcdr_dropout_trains_and_is_identity_in_eval():
    from oracle.common import IdSpace
    from recbole_cdr_amd.model.cross_domain_recommender.dtcdr import DTCDR
    ids = IdSpace(12, 10, 14, 1, 20, 24)
    cfg = base_config(DEV, embedding_size=16, mlp_hidden_size=[32, 16], dropout_prob=0.5, base_model='NeuMF', alpha=0.5)
    torch.manual_seed(0)
    model = DTCDR(cfg, FakeDataset(ids)).to(DEV)
    u = torch.randint(1, 12, (64,), device=DEV); i = torch.randint(1, 21, (64,), device=DEV)
    model.eval()
    a, b = model.neumf_forward(u, i, 'target'), model.neumf_forward(u, i, 'target')
    assert torch.equal(a, b)
    model.train()
    c, d = model.neumf_forward(u, i, 'target'), model.neumf_forward(u, i, 'target')
    assert not torch.equal(c, d) and torch.isfinite(c).all()
    c.sum().backward()
    assert all(torch.isfinite(p.grad).all() for p in model.parameters() if p.grad is not None)


def test_dtcdr_training_dropout_value_parity_with_the_same_mask():
    """DTCDR with dropout_prob > 0 in training mode: VALUES, not only statistics.  The product's counter-based mask of every (domain,
    layer) is exported with cdr_dropout_dev on a tensor of ones -- same device seed, same salt (source: layer index, target: 64 + layer
    index), same element order -- and handed to the oracle's MLP: loss and every gradient at 1e-5."""
    from oracle import dtcdr as o_dt
    from oracle.common import IdSpace
    from recbole_cdr_amd import binding as B_
    from recbole_cdr_amd.model.cross_domain_recommender.dtcdr import DTCDR
    ids = IdSpace(12, 10, 14, 1, 20, 24)
    D, hidden, p, B = 16, [32, 16], 0.4, 96
    cfg = base_config(DEV, embedding_size=D, mlp_hidden_size=hidden, dropout_prob=p, base_model='NeuMF', alpha=0.3)
    torch.manual_seed(3)
    model = DTCDR(cfg, FakeDataset(ids)).to(DEV)
    model.train()
    rng = np.random.RandomState(5)
    inter = {'source_user_id': torch.from_numpy(rng.randint(1, ids.total_num_users, B)), 'source_item_id': torch.from_numpy(rng.randint(1, ids.total_num_items, B)),
             'source_label': torch.from_numpy((rng.rand(B) < 0.5).astype(np.float32)),
             'target_user_id': torch.from_numpy(rng.randint(1, ids.OU + ids.TOU, B)), 'target_item_id': torch.from_numpy(rng.randint(1, ids.OI + ids.TOI, B)),
             'target_label': torch.from_numpy((rng.rand(B) < 0.5).astype(np.float32))}
    # the seed the model will draw for this forward (DTCDR._drop_seed: one draw from torch's CPU generator per eager training forward)
    torch.manual_seed(91)
    seed_val = int(torch.empty((), dtype=torch.int64).random_(0, 2 ** 62).item())
    seed = torch.full((1,), seed_val, device=DEV, dtype=torch.int64)
    masks = {}
    widths = [2 * D] + hidden[:-1]
    for dom, salt0 in (('source', 0), ('target', 64)):
        for n, w in enumerate(widths):
            ones = torch.ones(B, w, device=DEV)
            m = torch.empty_like(ones)
            B_.call('cdr_dropout_dev', B_.stream(), B_.f32(ones), ones.numel(), p, B_.i64(seed), salt0 + n, B_.f32(m))
            kept = float((m != 0).float().mean())
            assert abs(kept - (1 - p)) < 0.08 and bool(((m == 0) | ((m - 1 / (1 - p)).abs() < 1e-6)).all())
            masks[(dom, n)] = m.cpu()
    params = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in model.named_parameters()}
    want = o_dt.calculate_loss(params, ids, inter, 0.3, masks)
    want.backward()
    torch.manual_seed(91)
    loss = model.calculate_loss(to_dev(inter, DEV))
    assert_close(loss, want, what='loss')
    loss.backward()
    for k, v in model.named_parameters():
        if params[k].grad is not None:
            assert_close(v.grad, params[k].grad, what=k)


@pytest.mark.parametrize('name', cases('deepapf_'))
def test_deepapf_golden(name):
    from recbole_cdr_amd.model.cross_domain_recommender.deepapf import DeepAPF
    g = Golden(name)
    ids, ds = _ds(g)
    model = DeepAPF(base_config(DEV, embedding_size=int(g.meta('D')), beta=0.5), ds).to(DEV)
    assert {n for n, _ in model.named_parameters()} == set(g.group('param'))      # user_mlp.* and seq.* (deepapf.py:54-60)
    load_params(model, g.group('param'))
    inter = to_dev(g.group('in'), DEV)
    loss = model.calculate_loss(inter)
    assert_close(loss, g['loss/BOTH'], what=f'{name}:loss')
    loss.backward()
    _check_grads(model, g, 'BOTH')
    with torch.no_grad():
        assert_close(model.source_forward(inter['source_user_id'], inter['source_item_id']), g['fwd/source'], what='fwd/source')
        assert_close(model.target_forward(inter['target_user_id'], inter['target_item_id']), g['fwd/target'], what='fwd/target')
    assert_close(model.predict(to_dev(g.group('evalin'), DEV)), g['predict/BOTH'], what='predict')


@pytest.mark.parametrize('name', cases('natr_'))
def test_natr_golden(name):
    from recbole_cdr_amd.model.cross_domain_recommender.natr import NATR
    g = Golden(name)
    ids, ds = _ds(g, pairs=True)
    cfg = base_config(DEV, source_embedding_size=int(g.meta('Ds')), target_embedding_size=int(g.meta('Dt')),
                      reg_weight=float(g.meta('reg_weight')), max_inter_length=int(g.meta('max_inter_length')))
    model = NATR(cfg, ds).to(DEV)
    hist = model.history_user_matrix if model.mode == 'overlap_users' else model.history_item_matrix
    assert hist.device.type == 'cuda'
    np.testing.assert_array_equal(hist.cpu().numpy(), g['aux/history_matrix'])          # dataset.py:181-249, bit-exact
    np.testing.assert_array_equal(model.history_lens.cpu().numpy(), g['aux/history_lens'])
    np.testing.assert_array_equal(model.mask_mat.cpu().numpy(), g['aux/mask_mat'])
    load_params(model, g.group('param'))
    inter = to_dev(g.group('in'), DEV)
    ev = to_dev(g.group('evalin'), DEV)
    for phase in ('SOURCE', 'TARGET'):
        model.set_phase(phase)
        model.zero_grad(set_to_none=True)
        loss = model.calculate_loss(inter)
        assert_close(loss, g[f'loss/{phase}'], what=f'{name}:{phase}:loss')
        loss.backward()
        want = g.group(f'grad/{phase}', as_torch=False)
        got = {n: p.grad for n, p in model.named_parameters() if p.grad is not None}
        for n_, ref in want.items():
            if n_.endswith('attention_layer.bias'):
                # both attention biases cancel out of the loss (a softmax and a ratio e^a / (e^a + e^b) are shift-invariant): the
                # reference's value is rounding residue of a sum that is 0 in exact arithmetic; bound it by the gate's scale
                assert abs(float(got[n_].reshape(-1)[0]) - float(np.asarray(ref).reshape(-1)[0])) <= 1e-6, n_
            else:
                assert_close(got[n_], ref, what=f'{name}:{phase}:{n_}', row_floor=1e-2)
        assert set(got) == set(want), (sorted(got), sorted(want))
        assert_close(model.predict(ev), g[f'predict/{phase}'], what=f'{phase}:predict')
    model.set_phase('BOTH')
    assert model.calculate_loss(inter) is None


@pytest.mark.parametrize('name', cases('dcdcsr_'))
def test_dcdcsr_golden(name):
    from recbole_cdr_amd.model.cross_domain_recommender.dcdcsr import DCDCSR
    g = Golden(name)
    ids, ds = _ds(g, pairs=True)
    cfg = base_config(DEV, latent_factor_model='BPR', embedding_size=int(g.meta('D')),
                      mlp_hidden_size=[int(x) for x in g.meta('mlp_hidden_size')], k=int(g.meta('k')),
                      map_batch_size=int(g.meta('map_batch_size')))
    model = DCDCSR(cfg, ds).to(DEV)
    unit = 'user' if model.mode == 'overlap_users' else 'item'
    np.testing.assert_array_equal(getattr(model, f'source_{unit}2pop').cpu().numpy(), g['aux/source_pop'])
    np.testing.assert_array_equal(getattr(model, f'target_{unit}2pop').cpu().numpy(), g['aux/target_pop'])
    load_params(model, g.group('param'))
    inter = to_dev(g.group('in'), DEV)
    ev = to_dev(g.group('evalin'), DEV)

    def step(phase, tag):
        model.set_phase(phase)
        model.zero_grad(set_to_none=True)
        if phase == 'BOTH':
            assert_close(model.benchmark_embedding, g['fwd/benchmark_embedding'], what='benchmark')
            np.random.seed(77)
        if tag == 'TARGET2':
            assert_close(model.affine_embedding, g['fwd/affine_embedding'], what='affine')
        loss = model.calculate_loss(inter)
        assert_close(loss, g[f'loss/{tag}'], what=f'{name}:{tag}:loss')
        loss.backward()
        _check_grads(model, g, tag)
        if phase != 'BOTH':
            assert_close(model.predict(ev), g[f'predict/{tag}'], what=f'{tag}:predict')
            fs = model.full_sort_predict(ev)
            assert tuple(fs.shape) == tuple(g[f'fullsort/{tag}'].shape)                  # [U, N], not flattened (dcdcsr.py:244)
            assert_close(fs, g[f'fullsort/{tag}'], what=f'{tag}:fullsort')
    step('SOURCE', 'SOURCE'); step('TARGET', 'TARGET'); step('BOTH', 'BOTH'); step('TARGET', 'TARGET2')


# ---------------------------------------------------------------------------------------------- kernels vs the oracle at default sizes
def test_natr_attention_vs_oracle_default_size():
    """D = 64, max_inter_length = 50, batch 2,048 (properties/model/NATR.yaml) against the oracle's torch restatement."""
    from oracle import natr as o_natr
    from oracle.common import IdSpace
    from recbole_cdr_amd import functional as F_, binding as B_
    torch.manual_seed(3)
    Bn, L, Ds, Dt = 2048, 50, 64, 64
    ids = IdSpace(1, 300, 0, 200, 400, 500)          # overlap_items: the history belongs to the user
    nu, ni = ids.total_num_users, ids.total_num_items
    P = {'target_user_embedding.weight': torch.randn(nu, Dt) * 0.3, 'target_item_embedding.weight': torch.randn(ni, Dt) * 0.3,
         'source_item_embedding.weight': torch.randn(ni, Ds) * 0.3, 'source_user_embedding.weight': torch.randn(nu, Ds) * 0.3,
         'transfer_layer.weight': torch.randn(Dt, Ds) * 0.2, 'transfer_layer.bias': torch.randn(Dt) * 0.1,
         'unit_attention_layer.weight': torch.randn(1, Dt) * 0.3, 'unit_attention_layer.bias': torch.randn(1) * 0.1,
         'domain_attention_layer.weight': torch.randn(1, Dt) * 0.3, 'domain_attention_layer.bias': torch.randn(1) * 0.1}
    for k in P:
        P[k].requires_grad_(not k.startswith('source_'))
    lens = torch.randint(0, L + 20, (nu,))
    mat = torch.randint(1, ni, (nu, L))
    mat = mat * (torch.arange(L) < lens.unsqueeze(1))
    mask = (torch.arange(L) < lens.unsqueeze(1)).float()
    user, item = torch.randint(0, nu, (Bn,)), torch.randint(0, ni, (Bn,))
    label = (torch.rand(Bn) < 0.3).float()
    ref_p = o_natr.phase2_forward(P, ids, (mat, lens, mask), user, item)
    ref = torch.nn.functional.binary_cross_entropy(ref_p, label)
    ref.backward()
    d = {k: v.detach().to(DEV).requires_grad_(v.requires_grad) for k, v in P.items()}
    ue = F_.gather_rows(d['target_user_embedding.weight'], user.to(DEV))
    ie = F_.gather_rows(d['target_item_embedding.weight'], item.to(DEV))
    he = F_.linear(F_.gather_rows(d['source_item_embedding.weight'], mat.to(DEV)[user.to(DEV)]), d['transfer_layer.weight'],
                   d['transfer_layer.bias'], B_.ACT_NONE)
    p = F_.NatrAttention.apply(he, ue, ie, mask.to(DEV)[user.to(DEV)], d['unit_attention_layer.weight'],
                               d['unit_attention_layer.bias'], d['domain_attention_layer.weight'], d['domain_attention_layer.bias'])
    assert_close(p, ref_p, what='scores')
    loss = F_.BCEProbLoss.apply(p, label.to(DEV))
    assert_close(loss, ref, what='loss')
    loss.backward()
    for k in P:
        if not P[k].requires_grad:
            continue
        if k.endswith('attention_layer.bias'):
            assert abs(float(d[k].grad) - float(P[k].grad)) <= 1e-6
        else:
            assert_close(d[k].grad, P[k].grad, what=k, row_floor=1e-2)


@pytest.mark.parametrize('D', [8, 64, 100])
def test_maxmin_normalize_vs_oracle(D):
    from oracle import dcdcsr as o
    from recbole_cdr_amd import functional as F_
    torch.manual_seed(D)
    x = torch.randn(513, D)
    x[5, 1] = x[5].max(); x[5, 2] = x[5].max()          # tied maxima
    x[6, 0] = x[6].min(); x[6, 3] = x[6].min()          # tied minima
    x.requires_grad_(True)
    w = torch.randn(513, D)
    y, mean_, max_ = o.maxmin_normalize(x)
    (y * w).sum().backward()
    xd = x.detach().to(DEV).requires_grad_(True)
    yd, stats = F_.MaxMinNormalize.apply(xd)
    assert_close(yd, y, what='y')
    assert_close(stats[:, 0], mean_.reshape(-1), what='mean'); assert_close(stats[:, 1], max_.reshape(-1), what='max')
    (yd * w.to(DEV)).sum().backward()
    assert_close(xd.grad, x.grad, what='gx', row_floor=1e-2)


def test_deepapf_default_size_vs_oracle():
    """D = 64, batch 2,048 with masked and unmasked rows against the oracle's torch restatement."""
    from oracle import deepapf as o
    from oracle.common import IdSpace
    from recbole_cdr_amd.model.cross_domain_recommender.deepapf import DeepAPF
    ids = IdSpace(120, 100, 140, 1, 200, 240)
    torch.manual_seed(5)
    model = DeepAPF(base_config(DEV, embedding_size=64, beta=0.5), FakeDataset(ids)).to(DEV)
    with torch.no_grad():
        for n, p in model.named_parameters():
            if n.endswith('bias'):
                p.normal_(0, 0.1)
    P = {n: p.detach().cpu().clone().requires_grad_(True) for n, p in model.named_parameters()}
    Bn = 2048
    inter = {'source_user_id': torch.randint(0, ids.total_num_users, (Bn,)), 'source_item_id': torch.randint(0, ids.total_num_items, (Bn,)),
             'target_user_id': torch.randint(0, ids.target_num_users, (Bn,)), 'target_item_id': torch.randint(0, ids.target_num_items, (Bn,)),
             'source_label': (torch.rand(Bn) < 0.3).float(), 'target_label': (torch.rand(Bn) < 0.3).float()}
    ref = o.calculate_loss(P, ids, inter)
    ref.backward()
    loss = model.calculate_loss(to_dev(inter, DEV))
    assert_close(loss, ref, what='loss')
    loss.backward()
    for n, p in model.named_parameters():
        if P[n].grad is None:
            assert p.grad is None or float(p.grad.abs().max()) == 0.0
        else:
            assert_close(p.grad, P[n].grad, what=n, row_floor=1e-2)


# ---------------------------------------------------------------------------------------------- through the trainers
def _loaders(ids, input_type, k, seed, n_s=96, n_t=80, bs=32, OB=8):
    from recbole_cdr_amd.data import CrossDomainDataloader, OverlapDataloader, DomainTrainLoader
    rng = np.random.RandomState(seed)
    src_u = np.array(list(range(1, ids.OU)) + list(range(ids.OU + ids.TOU, ids.total_num_users)))
    src_i = np.array(list(range(1, ids.OI)) + list(range(ids.OI + ids.TOI, ids.total_num_items)))
    tgt_u, tgt_i = np.arange(1, ids.OU + ids.TOU), np.arange(1, ids.OI + ids.TOI)
    s_pairs = np.stack([np.concatenate([src_u, rng.choice(src_u, n_s)]), np.concatenate([rng.choice(src_i, len(src_u)), rng.choice(src_i, n_s)])], 1)
    t_pairs = np.stack([np.concatenate([tgt_u, rng.choice(tgt_u, n_t)]), np.concatenate([rng.choice(tgt_i, len(tgt_u)), rng.choice(tgt_i, n_t)])], 1)
    # every item appears as well (popularities > 0: dcdcsr.py:147-149 divides by them)
    s_pairs = np.concatenate([s_pairs, np.stack([rng.choice(src_u, len(src_i)), src_i], 1)])
    t_pairs = np.concatenate([t_pairs, np.stack([rng.choice(tgt_u, len(tgt_i)), tgt_i], 1)])
    s_inter = {'source_user_id': torch.from_numpy(s_pairs[:, 0].copy()), 'source_item_id': torch.from_numpy(s_pairs[:, 1].copy())}
    t_inter = {'target_user_id': torch.from_numpy(t_pairs[:, 0].copy()), 'target_item_id': torch.from_numpy(t_pairs[:, 1].copy())}
    neg_rng = {}

    def reset():
        neg_rng['s'], neg_rng['t'] = np.random.RandomState(seed + 1), np.random.RandomState(seed + 2)
    reset()
    s_sampler = lambda u, i, kk: torch.from_numpy(neg_rng['s'].choice(src_i, u.numel() * kk)).to(u.device)
    t_sampler = lambda u, i, kk: torch.from_numpy(neg_rng['t'].choice(tgt_i, u.numel() * kk)).to(u.device)
    mk = lambda: CrossDomainDataloader(
        DomainTrainLoader(s_inter, 'source_user_id', 'source_item_id', 'source_label', 'neg_', bs, k, input_type, s_sampler),
        DomainTrainLoader(t_inter, 'target_user_id', 'target_item_id', 'target_label', 'neg_', bs, k, input_type, t_sampler),
        OverlapDataloader(max(ids.OU, ids.OI), OB))
    return mk, reset, s_pairs, t_pairs


def _epoch_batches(dl, phase):
    from recbole_cdr_amd.utils import train_mode2state
    dl.set_mode(train_mode2state[phase])
    it = iter(dl)
    while True:
        try:
            yield next(it)
        except StopIteration:
            return


def test_dcdcsr_trainer_phase_loop_matches_oracle_training():
    """DCDCSRTrainer.fit over SOURCE -> TARGET -> BOTH -> TARGET (trainer.py:79-139; dense native Adam, one optimizer across the
    phases) against the oracle trained with torch.optim.Adam on the same batches and the same numpy draws: per-epoch loss sums,
    the benchmark / affine tables built at the phase switches, and the final parameters."""
    from oracle import dcdcsr as o
    from oracle.common import IdSpace
    from recbole_cdr_amd.model.cross_domain_recommender.dcdcsr import DCDCSR
    from recbole_cdr_amd.trainer import DCDCSRTrainer
    from recbole_cdr_amd.utils import InputType, get_trainer, ModelType
    assert get_trainer(ModelType.CROSSDOMAIN, 'DCDCSR') is DCDCSRTrainer
    torch.manual_seed(13)
    ids = IdSpace(OU=20, TOU=15, SOU=18, OI=1, TOI=30, SOI=34)
    mk, reset, s_pairs, t_pairs = _loaders(ids, InputType.PAIRWISE, 1, seed=4)
    modes, epochs = ['SOURCE', 'TARGET', 'BOTH', 'TARGET'], [2, 1, 2, 1]
    cfg = base_config(DEV, latent_factor_model='BPR', embedding_size=16, mlp_hidden_size=[24], k=4, map_batch_size=32,
                      learning_rate=0.01, train_modes=modes, epoch_num=[str(e) for e in epochs], source_split=False,
                      eval_step=1, epochs=2)
    ds = FakeDataset(ids, s_pairs, t_pairs)
    ds.device = DEV
    model = DCDCSR(cfg, ds).to(DEV)
    params = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in model.named_parameters()}
    trainer = DCDCSRTrainer(cfg, model)
    log, orig = [], trainer._train_epoch
    trainer._train_epoch = lambda data, e: (log.append(orig(data, e)) or log[-1])
    np.random.seed(5)
    trainer.fit(mk())
    assert model.phase == 'OVERLAP' and len(log) == sum(epochs)
    # ---- oracle
    reset()
    np.random.seed(5)
    pops = o.unit_pops(ids, s_pairs, t_pairs)
    opt = torch.optim.Adam(list(params.values()), lr=0.01)
    dl = mk()
    ref_log, seen, bench, affine = [], {}, None, None
    for phase, n_ep in zip(modes, epochs):
        seen[phase] = seen.get(phase, 0) + 1
        stage = 'TARGET2' if (phase == 'TARGET' and seen[phase] == 2) else phase
        if phase == 'BOTH':
            bench = o.build_benchmark_embedding(params, ids, pops, 4)
        if stage == 'TARGET2':
            affine = o.build_affine_embedding(params, ids)
        for _ in range(n_ep):
            tot = 0.0
            for b in _epoch_batches(dl, phase):
                opt.zero_grad()
                if phase == 'BOTH':
                    loss = o.map_loss(params, ids, bench, np.random.randint(0, ids.target_num_users, 32))
                else:
                    loss = o.rec_loss(params, ids, b, stage, affine)
                loss.backward()
                opt.step()
                tot += float(loss.detach())
            ref_log.append(tot)
    assert_close(torch.tensor(log), torch.tensor(ref_log), rtol=2e-5, what='epoch losses')
    assert_close(model.benchmark_embedding, bench, rtol=1e-4, atol=1e-5, what='benchmark table after training')
    assert_close(model.affine_embedding, affine, rtol=1e-4, atol=1e-4, what='affine table after training')
    for k, v in model.named_parameters():
        assert_close(v, params[k], rtol=1e-4, atol=0.01 * 2e-2, what=k)     # a few Adam steps ~ lr * sign(g): 2 % of one step


def test_natr_trainer_phase_loop_matches_oracle_training():
    """CrossDomainTrainer.fit over SOURCE -> TARGET with NATR (POINTWISE batches with labelled negatives; the TARGET phase freezes the
    source tables, natr.py:69-73) against the oracle trained with torch.optim.Adam on the same batches."""
    from oracle import natr as o
    from oracle.common import IdSpace
    from recbole_cdr_amd.model.cross_domain_recommender.natr import NATR
    from recbole_cdr_amd.trainer import CrossDomainTrainer
    from recbole_cdr_amd.utils import InputType
    torch.manual_seed(17)
    ids = IdSpace(OU=1, TOU=25, SOU=22, OI=18, TOI=20, SOI=16)
    mk, reset, s_pairs, t_pairs = _loaders(ids, InputType.POINTWISE, 1, seed=6)
    cfg = base_config(DEV, source_embedding_size=16, target_embedding_size=24, reg_weight=1e-3, max_inter_length=6,
                      learning_rate=0.01, train_modes=['SOURCE', 'TARGET'], epoch_num=['2', '2'], source_split=False,
                      eval_step=1, epochs=2)
    ds = FakeDataset(ids, s_pairs, t_pairs)
    ds.device = DEV
    model = NATR(cfg, ds).to(DEV)
    params = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in model.named_parameters()}
    trainer = CrossDomainTrainer(cfg, model)
    log, orig = [], trainer._train_epoch
    trainer._train_epoch = lambda data, e: (log.append(orig(data, e)) or log[-1])
    trainer.fit(mk())
    assert len(log) == 4
    reset()
    hist = o.history_info(ids, t_pairs, 6)
    opt = torch.optim.Adam(list(params.values()), lr=0.01)
    dl = mk()
    ref_log = []
    for phase in ('SOURCE', 'TARGET'):
        if phase == 'TARGET':
            params['source_user_embedding.weight'].requires_grad_(False)
            params['source_item_embedding.weight'].requires_grad_(False)
        for _ in range(2):
            tot = 0.0
            for b in _epoch_batches(dl, phase):
                opt.zero_grad()
                loss = o.calculate_loss(params, ids, hist, b, phase, 1e-3)
                loss.backward()
                opt.step()
                tot += float(loss.detach())
            ref_log.append(tot)
    assert_close(torch.tensor(log), torch.tensor(ref_log), rtol=2e-5, what='epoch losses')
    for k, v in model.named_parameters():
        if k.endswith('attention_layer.bias'):
            # the loss does not depend on either attention bias (softmax / gate shift invariance): their gradients are rounding residue
            # (~1e-9) that Adam normalises into steps of arbitrary sign -- both implementations drift, by less than lr per step
            assert abs(float(v.detach()) - float(params[k].detach())) <= 0.01 * 40
            continue
        assert_close(v, params[k], rtol=1e-4, atol=0.01 * 2e-2, what=k)


def test_five_models_replay_as_one_hipgraph_equal_to_eager():
    """graph_step.GraphedTrainStep on CLFM / DTCDR / DeepAPF / NATR phase 2 / DCDCSR's BPR phase: 3 replays on changing batches leave
    the same loss values as the eager loop from the same start (1e-5: the dense scatter-add's fp32 atomics reorder sums), and DTCDR
    with dropout draws a fresh mask on every replay (the seed is a device counter advanced inside the captured step)."""
    from oracle.common import IdSpace
    from recbole_cdr_amd.graph_step import GraphedTrainStep
    from recbole_cdr_amd.trainer.trainer import DenseAdam
    from recbole_cdr_amd.model.cross_domain_recommender.clfm import CLFM
    from recbole_cdr_amd.model.cross_domain_recommender.dtcdr import DTCDR
    from recbole_cdr_amd.model.cross_domain_recommender.deepapf import DeepAPF
    from recbole_cdr_amd.model.cross_domain_recommender.natr import NATR
    from recbole_cdr_amd.model.cross_domain_recommender.dcdcsr import DCDCSR
    ids = IdSpace(OU=1, TOU=60, SOU=70, OI=40, TOI=50, SOI=45)
    rng = np.random.RandomState(1)
    src_u = np.arange(ids.OU + ids.TOU, ids.total_num_users)
    src_i = np.r_[1:ids.OI, ids.OI + ids.TOI:ids.total_num_items]
    s_pairs = np.stack([rng.choice(src_u, 900), rng.choice(src_i, 900)], 1)
    t_pairs = np.stack([rng.randint(1, ids.OU + ids.TOU, 900), rng.randint(1, ids.OI + ids.TOI, 900)], 1)
    ds = FakeDataset(ids, s_pairs, t_pairs)
    ds.device = DEV

    def batch(pairwise, B=128):
        t = lambda a: torch.from_numpy(np.asarray(a, dtype=np.int64)).to(DEV)
        out = {}
        for d, us, its in (('source', src_u, src_i), ('target', np.arange(1, ids.OU + ids.TOU), np.arange(1, ids.OI + ids.TOI))):
            out[f'{d}_user_id'], out[f'{d}_item_id'] = t(rng.choice(us, B)), t(rng.choice(its, B))
            if pairwise:
                out[f'neg_{d}_item_id'] = t(rng.choice(its, B))
            else:
                out[f'{d}_label'] = torch.from_numpy((rng.rand(B) < 0.5).astype(np.float32)).to(DEV)
        return out
    cases = [(CLFM, dict(user_embedding_size=16, source_item_embedding_size=16, target_item_embedding_size=16, share_embedding_size=8,
                         alpha=0.5, reg_weight=1e-3), False, None),
             (DTCDR, dict(embedding_size=16, mlp_hidden_size=[16, 8], dropout_prob=0.0, base_model='NeuMF', alpha=0.5), False, None),
             (DeepAPF, dict(embedding_size=16, beta=0.5), False, None),
             (NATR, dict(source_embedding_size=16, target_embedding_size=16, reg_weight=1e-3, max_inter_length=8), False, 'TARGET'),
             (DCDCSR, dict(latent_factor_model='BPR', embedding_size=16, mlp_hidden_size=[16], k=3, map_batch_size=32), True, 'TARGET')]
    for cls, kw, pairwise, phase in cases:
        batches = [batch(pairwise) for _ in range(4)]
        runs = []
        for graphed in (False, True):
            torch.manual_seed(9)
            m = cls(base_config(DEV, **kw), ds).to(DEV)
            if phase:
                m.set_phase(phase)
            m.train()
            opt = DenseAdam(m.parameters(), lr=0.01)
            if graphed:
                g = GraphedTrainStep(m, opt, batches[0])
                runs.append([float(g.step(b)) for b in batches])
            else:
                out = []
                for b in batches:
                    opt.zero_grad(set_to_none=True)
                    loss = m.calculate_loss(b).sum()
                    loss.backward()
                    opt.step()
                    out.append(float(loss.detach()))
                runs.append(out)
        assert_close(torch.tensor(runs[1]), torch.tensor(runs[0]), rtol=1e-5, atol=1e-7, what=cls.__name__)
    torch.manual_seed(3)
    m = DTCDR(base_config(DEV, embedding_size=16, mlp_hidden_size=[16, 8], dropout_prob=0.5, base_model='NeuMF', alpha=0.5), ds).to(DEV)
    m.train()
    b = batch(False)
    g = GraphedTrainStep(m, DenseAdam(m.parameters(), lr=0.0), b, warmup=1)
    losses = [float(g.step(b)) for _ in range(4)]
    assert len({round(l, 7) for l in losses}) == 4, losses
