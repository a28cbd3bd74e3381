"""The oracle (oracle/) against golden vectors produced by the reference's own code (tests/golden/make_golden.py).

fp32 bar: 1e-6 relative (same torch ops in the same order -> normally bit-identical); integer paths bit-exact.
"""
import numpy as np
import pytest
import torch

import oracle
from oracle import emcdr, cmf, conet, sscdr, bitgcf, remap, clfm, dtcdr, deepapf, natr, dcdcsr, history
from golden_util import Golden, cases

RTOL, ATOL = 1e-6, 1e-7


def close(a, b, rtol=RTOL, atol=ATOL):
    a = a.detach().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
    np.testing.assert_allclose(a.reshape(-1), np.asarray(b).reshape(-1), rtol=rtol, atol=atol)


def check_grads(params, g, phase):
    want = g.group(f'grad/{phase}', as_torch=False)
    assert want, f'no grads recorded for {phase}'
    for name, ref in want.items():
        got = params[name].grad
        assert got is not None, name
        close(got, ref, rtol=1e-5, atol=1e-7)
    for name, p in params.items():
        if name not in want and p.grad is not None:
            assert float(p.grad.abs().max()) == 0.0, f'{name} has a gradient the reference does not produce'


def zero_grads(params):
    for p in params.values():
        p.grad = None


@pytest.mark.parametrize('name', cases('emcdr_'))
def test_emcdr(name):
    g = Golden(name)
    ids = g.idspace()
    params = g.group('param', requires_grad=True)
    inter = g.group('in')
    lfm = str(g.meta('latent_factor_model'))
    for phase in ('SOURCE', 'TARGET', 'OVERLAP', 'BOTH'):
        zero_grads(params)
        loss = emcdr.calculate_loss(params, ids, inter, phase, lfm, float(g.meta('reg_weight')))
        close(loss, g[f'loss/{phase}'])
        loss.sum().backward()
        check_grads(params, g, phase)
    ev = g.group('evalin')
    with torch.no_grad():
        for phase in ('SOURCE', 'TARGET', 'OVERLAP', 'BOTH'):
            close(emcdr.predict(params, ids, ev, phase), g[f'predict/{phase}'])
            close(emcdr.full_sort_predict(params, ids, ev, phase), g[f'fullsort/{phase}'])


@pytest.mark.parametrize('name', cases('cmf_'))
def test_cmf(name):
    g = Golden(name)
    ids = g.idspace()
    params = g.group('param', requires_grad=True)
    inter = g.group('in')
    loss = cmf.calculate_loss(params, ids, inter, float(g.meta('alpha')), float(g.meta('lam')), float(g.meta('gamma')))
    close(loss, g['loss/BOTH'])
    loss.sum().backward()
    check_grads(params, g, 'BOTH')
    ev = g.group('evalin')
    with torch.no_grad():
        close(cmf.predict(params, ids, ev), g['predict/BOTH'])
        close(cmf.full_sort_predict(params, ids, ev), g['fullsort/BOTH'])


@pytest.mark.parametrize('name', cases('conet_'))
def test_conet(name):
    g = Golden(name)
    ids = g.idspace()
    params = g.group('param', requires_grad=True)
    inter = g.group('in')
    loss = conet.calculate_loss(params, ids, inter)
    close(loss, g['loss/BOTH'])
    loss.backward()
    check_grads(params, g, 'BOTH')
    with torch.no_grad():
        close(conet.source_forward(params, ids, inter['source_user_id'], inter['source_item_id']), g['fwd/source'])
        close(conet.target_forward(params, ids, inter['target_user_id'], inter['target_item_id']), g['fwd/target'])
        ev = g.group('evalin')
        p = conet.predict(params, ids, ev)
        assert tuple(p.shape) == tuple(g['predict/BOTH'].shape)          # [B,1]
        close(p, g['predict/BOTH'])
        fs = conet.full_sort_predict(params, ids, ev)
        assert tuple(fs.shape) == tuple(g['fullsort/BOTH'].shape)        # [U,N]
        close(fs, g['fullsort/BOTH'])


@pytest.mark.parametrize('name', cases('sscdr_'))
def test_sscdr(name):
    g = Golden(name)
    ids = g.idspace()
    params = g.group('param', requires_grad=True)
    inter = g.group('in')
    margin, lam = float(g.meta('margin')), float(g.meta('lam'))
    # the sampler restatement reproduces the reference's draws from the same numpy seed
    indptr, indices = g['aux/hist_indptr'], g['aux/hist_indices']
    hist = [list(indices[indptr[i]:indptr[i + 1]]) for i in range(len(indptr) - 1)]
    np.random.seed(99)
    mode = 'user' if ids.mode == 'overlap_users' else 'item'
    pos, neg = sscdr.sample(inter['overlap'].squeeze(1).numpy(), ids, hist, mode)
    np.testing.assert_array_equal(pos.numpy(), g['aux/sampled_pos'])
    np.testing.assert_array_equal(neg.numpy(), g['aux/sampled_neg'])
    for phase in ('SOURCE', 'TARGET', 'BOTH', 'OVERLAP'):
        zero_grads(params)
        loss = sscdr.calculate_loss(params, ids, inter, phase, margin, lam, pos, neg)
        close(loss, g[f'loss/{phase}'])
        loss.backward()
        check_grads(params, g, phase)
    ev = g.group('evalin')
    with torch.no_grad():
        for phase in ('SOURCE', 'TARGET', 'OVERLAP'):
            close(sscdr.predict(params, ids, ev, phase), g[f'predict/{phase}'], rtol=1e-5)
            close(sscdr.full_sort_predict(params, ids, ev, phase), g[f'fullsort/{phase}'], rtol=1e-5)


@pytest.mark.parametrize('name', cases('bitgcf_'))
def test_bitgcf(name):
    g = Golden(name)
    ids = g.idspace()
    params = g.group('param', requires_grad=True)
    inter = g.group('in')
    graph = bitgcf.build_graph(g['aux/s_pairs'], g['aux/t_pairs'], ids.total_num_users, ids.total_num_items)
    for dom in ('source', 'target'):
        a = graph[f'{dom}_adj']
        np.testing.assert_array_equal(a.indices().numpy(), g[f'aux/adj_{dom}_idx'])
        np.testing.assert_array_equal(a.values().numpy(), g[f'aux/adj_{dom}_val'])     # bit-exact fp32
    kw = dict(n_layers=int(g.meta('n_layers')), lam_s=float(g.meta('lambda_source')),
              lam_t=float(g.meta('lambda_target')), connect_way=str(g.meta('connect_way')))
    losses = bitgcf.calculate_loss(params, ids, graph, inter, reg_weight=float(g.meta('reg_weight')), **kw)
    close(torch.stack([l.reshape(()) for l in losses]), g['loss/BOTH'])
    sum(losses).sum().backward()
    check_grads(params, g, 'BOTH')
    with torch.no_grad():
        su, si, tu, ti = bitgcf.forward(params, ids, graph, **kw)
        close(su, g['fwd/source_user']); close(si, g['fwd/source_item'])
        close(tu, g['fwd/target_user']); close(ti, g['fwd/target_item'])
        ev = g.group('evalin')
        close(bitgcf.predict(params, ids, graph, ev, **kw), g['predict/BOTH'])
        close(bitgcf.full_sort_predict(params, ids, graph, ev, **kw), g['fullsort/BOTH'])


def _tokens(g, key):
    toks = [str(t) for t in g[f'in/{key}_tokens']]
    nan = g[f'in/{key}_isnan'] if g.has(f'in/{key}_isnan') else np.zeros(len(toks), bool)
    return [None if m else t for t, m in zip(toks, nan)]


@pytest.mark.parametrize('name', cases('remap_'))
def test_remap_bit_exact(name):
    g = Golden(name)
    su, si, tu, ti = (_tokens(g, k) for k in ('source_user', 'source_item', 'target_user', 'target_item'))
    su_all = su + ([str(t) for t in g['in/source_user_feat_tokens']] if g.has('in/source_user_feat_tokens') else [])
    tu_all = tu + ([str(t) for t in g['in/target_user_feat_tokens']] if g.has('in/target_user_feat_tokens') else [])
    msu, msi, mtu, mti, counts = remap.overlap_remap(su_all, si, tu_all, ti)
    for prefix, m in (('source_user', msu), ('source_item', msi), ('target_user', mtu), ('target_item', mti)):
        want = dict(zip((str(t) for t in g[f'{prefix}/tokens']), g[f'{prefix}/ids'].tolist()))
        assert m == want, prefix
    for k in ('num_overlap_user', 'num_source_only_user', 'num_target_only_user', 'num_total_user',
              'num_overlap_item', 'num_source_only_item', 'num_target_only_item', 'num_total_item'):
        assert counts[k] == int(g[f'count/{k}']), k
    np.testing.assert_array_equal(remap.apply_remap(su, msu), g['applied/source_user'])
    np.testing.assert_array_equal(remap.apply_remap(si, msi), g['applied/source_item'])
    np.testing.assert_array_equal(remap.apply_remap(tu, mtu), g['applied/target_user'])
    np.testing.assert_array_equal(remap.apply_remap(ti, mti), g['applied/target_item'])


def test_revoke_map_and_layout():
    g = Golden('revoke_layout')
    OI, TOI = int(g['revoke/OI']), int(g['revoke/TOI'])
    used_ptr, used = g['revoke/used_ptr'], g['revoke/used_flat']
    pos_ptr, pos = g['revoke/pos_ptr'], g['revoke/pos_flat']
    for uid in (1, 2, 4):
        u = set(used[used_ptr[uid]:used_ptr[uid + 1]].tolist())
        p = set(pos[pos_ptr[uid]:pos_ptr[uid + 1]].tolist())
        np.testing.assert_array_equal(np.sort(remap.revoke_map(sorted(p), OI, TOI)), g[f'revoke/positive/{uid}'])
        np.testing.assert_array_equal(np.sort(remap.revoke_map(sorted(u - p), OI, TOI)), g[f'revoke/history/{uid}'])
    sched = remap.BothModeSchedule(n_source=2, n_target=5, n_overlap=3)
    for state in ('BOTH', 'SOURCE', 'TARGET', 'OVERLAP'):
        assert sched.length(state) == int(g[f'layout/{state}/len'])
        trace = g[f'layout/{state}/trace']
        got = np.array(sched.epoch(state), dtype=np.int64)
        np.testing.assert_array_equal(got, trace[:, :3])
        np.testing.assert_array_equal(g[f'layout/{state}/pr_after'], [0, 0, 0])
    users, items = remap.source_id_lists(12, 10, 14, 1, 20, 24)
    assert users[0] == 1 and users[10] == 11 and users[11] == 22 and len(users) == 11 + 14
    assert items[0] == 21 and len(items) == 24


# ---------------------------------------------------------------------------------------------- the five remaining models
@pytest.mark.parametrize('name', cases('clfm_'))
def test_clfm(name):
    g = Golden(name)
    ids = g.idspace()
    params = g.group('param', requires_grad=True)
    inter = g.group('in')
    loss = clfm.calculate_loss(params, ids, inter, float(g.meta('alpha')), float(g.meta('reg_weight')))
    close(loss, g['loss/BOTH'])
    loss.sum().backward()
    check_grads(params, g, 'BOTH')
    ev = g.group('evalin')
    with torch.no_grad():
        close(clfm.predict(params, ids, ev), g['predict/BOTH'])
        close(clfm.full_sort_predict(params, ids, ev), g['fullsort/BOTH'])


@pytest.mark.parametrize('name', cases('dtcdr_'))
def test_dtcdr(name):
    g = Golden(name)
    ids = g.idspace()
    params = g.group('param', requires_grad=True)
    loss = dtcdr.calculate_loss(params, ids, g.group('in'), float(g.meta('alpha')))
    close(loss, g['loss/BOTH'])
    loss.backward()
    check_grads(params, g, 'BOTH')
    with torch.no_grad():
        close(dtcdr.predict(params, ids, g.group('evalin')), g['predict/BOTH'])


@pytest.mark.parametrize('name', cases('deepapf_'))
def test_deepapf(name):
    g = Golden(name)
    ids = g.idspace()
    params = g.group('param', requires_grad=True)
    inter = g.group('in')
    loss = deepapf.calculate_loss(params, ids, inter)
    close(loss, g['loss/BOTH'])
    loss.backward()
    check_grads(params, g, 'BOTH')
    with torch.no_grad():
        close(deepapf.forward(params, ids, inter['source_user_id'], inter['source_item_id'], 'source'), g['fwd/source'])
        close(deepapf.forward(params, ids, inter['target_user_id'], inter['target_item_id'], 'target'), g['fwd/target'])
        close(deepapf.predict(params, ids, g.group('evalin')), g['predict/BOTH'])


@pytest.mark.parametrize('name', cases('natr_'))
def test_natr(name):
    g = Golden(name)
    ids = g.idspace()
    params = g.group('param', requires_grad=True)
    inter = g.group('in')
    hist = natr.history_info(ids, g['aux/t_pairs'], int(g.meta('max_inter_length')))
    np.testing.assert_array_equal(hist[0].numpy(), g['aux/history_matrix'])            # dataset.py:181-249, bit-exact
    np.testing.assert_array_equal(hist[1].numpy(), g['aux/history_lens'])
    np.testing.assert_array_equal(hist[2].numpy(), g['aux/mask_mat'])
    ev = g.group('evalin')
    for phase in ('SOURCE', 'TARGET'):
        zero_grads(params)
        if phase == 'TARGET':                          # natr.py:69-73
            params['source_user_embedding.weight'].requires_grad_(False)
            params['source_item_embedding.weight'].requires_grad_(False)
        loss = natr.calculate_loss(params, ids, hist, inter, phase, float(g.meta('reg_weight')))
        close(loss, g[f'loss/{phase}'])
        loss.backward()
        check_grads(params, g, phase)
        with torch.no_grad():
            close(natr.predict(params, ids, hist, ev, phase), g[f'predict/{phase}'])
    assert natr.calculate_loss(params, ids, hist, inter, 'BOTH', 0.0) is None          # natr.py:175-176


@pytest.mark.parametrize('name', cases('dcdcsr_'))
def test_dcdcsr(name):
    g = Golden(name)
    ids = g.idspace()
    params = g.group('param', requires_grad=True)
    inter = g.group('in')
    ev = g.group('evalin')
    pops = dcdcsr.unit_pops(ids, g['aux/s_pairs'], g['aux/t_pairs'])
    np.testing.assert_array_equal(pops[0].numpy(), g['aux/source_pop'])
    np.testing.assert_array_equal(pops[1].numpy(), g['aux/target_pop'])

    def evals(stage, affine=None):
        with torch.no_grad():
            close(dcdcsr.predict(params, ids, ev, stage, affine), g[f'predict/{stage}'])
            close(dcdcsr.full_sort_predict(params, ids, ev, stage, affine), g[f'fullsort/{stage}'])
    for stage in ('SOURCE', 'TARGET'):
        zero_grads(params)
        loss = dcdcsr.rec_loss(params, ids, inter, stage)
        close(loss, g[f'loss/{stage}'])
        loss.backward()
        check_grads(params, g, stage)
        evals(stage)
    bench = dcdcsr.build_benchmark_embedding(params, ids, pops, int(g.meta('k')))
    close(bench, g['fwd/benchmark_embedding'])
    np.random.seed(77)
    unit_n = ids.target_num_users if ids.mode == 'overlap_users' else ids.target_num_items
    sampled = np.random.randint(0, unit_n, int(g.meta('map_batch_size')))                # dcdcsr.py:175
    np.testing.assert_array_equal(sampled, g['aux/sampled_index'])
    zero_grads(params)
    loss = dcdcsr.map_loss(params, ids, bench, sampled)
    close(loss, g['loss/BOTH'])
    loss.backward()
    check_grads(params, g, 'BOTH')
    affine = dcdcsr.build_affine_embedding(params, ids)
    close(affine, g['fwd/affine_embedding'])
    zero_grads(params)
    loss = dcdcsr.rec_loss(params, ids, inter, 'TARGET2', affine)
    close(loss, g['loss/TARGET2'])
    loss.backward()
    check_grads(params, g, 'TARGET2')
    evals('TARGET2', affine)
