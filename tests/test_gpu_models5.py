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
