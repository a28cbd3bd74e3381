"""Ids without a sort (round 6): the count path of the fused BPR step (csrc/cdr_step.hip, cdr_ctx_set_id_counters) against the sorted path on the
same batches -- flags, duplicate segments and every table / moment BIT-equal (same kernels behind identical operands), the counters left all
zero, the slow global-memory sort of a long duplicate list, the host binding's move to the sorted path on a skewed stream, graph replays."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def _pair(nu, ni, D, B, seed=0, **kw):
    from recbole_cdr_amd.fused import FusedBPRStep
    torch.manual_seed(seed)
    U0, I0 = torch.randn(nu, D, device=DEV) * 0.1, torch.randn(ni, D, device=DEV) * 0.1
    a = FusedBPRStep(U0.clone(), I0.clone(), B, opt='adam', lr=0.01, reg_weight=0.02, id_path='sort', **kw)
    b = FusedBPRStep(U0.clone(), I0.clone(), B, opt='adam', lr=0.01, reg_weight=0.02, id_path='count', **kw)
    return a, b


def _same(a, b):
    for x, y, what in ((a.U, b.U, 'U'), (a.I, b.I, 'I'), (a.ustate.exp_avg, b.ustate.exp_avg, 'mU'), (a.ustate.exp_avg_sq, b.ustate.exp_avg_sq, 'vU'),
                       (a.istate.exp_avg, b.istate.exp_avg, 'mI'), (a.istate.exp_avg_sq, b.istate.exp_avg_sq, 'vI')):
        assert torch.equal(x, y), what
    assert torch.equal(a.out6[:9], b.out6[:9])


@pytest.mark.parametrize('shape', ['uniform', 'dups', 'hot', 'long_list'])
def test_count_path_is_bit_equal_to_the_sorted_path(shape):
    nu, ni, D, B = 400000, 150000, 128, 40000
    if shape == 'uniform':
        nu, ni, D = 6000000, 4000000, 64                        # a few hundred duplicate occurrences: the list is sorted in LDS
    if shape == 'long_list':
        nu, ni = 30000, 20000                                   # ~ 70 % of the occurrences are duplicates: the list is sorted in global memory
    a, b = _pair(nu, ni, D, B)
    g = torch.Generator(device=DEV); g.manual_seed(5)
    for step in range(3):
        u = torch.randint(0, nu, (B,), device=DEV, generator=g)
        p = torch.randint(0, ni, (B,), device=DEV, generator=g)
        n = torch.randint(0, ni, (B,), device=DEV, generator=g)
        if shape == 'dups':
            u[:300] = u[0]; p[1000:1040] = p[7]; n[5:9] = p[7]     # a long user segment (pieces), medium item segments
        if shape == 'hot':
            p[: B // 8] = 11; n[B // 2: B // 2 + 700] = 11; u[100:140] = 3
        a.step(u, p, n); b.step(u, p, n)
        torch.cuda.synchronize()
        assert int(b._count[0].abs().sum()) == 0 and int(b._count[1].abs().sum()) == 0, 'counters must be left all zero'
        assert torch.equal(a.flags[:4 * B].view(torch.int32) & 0x010101, b.flags[:4 * B].view(torch.int32) & 0x010101)
        assert torch.equal(a.heads[:2], b.heads[:2]) and int(a.heads[2]) == int(b.heads[2])      # segment counts, duplicate occurrences
        _same(a, b)
    nd = int(b.heads[2])
    if shape == 'long_list':
        assert nd > 16384
    if shape == 'uniform':
        assert 0 < nd < 16384


def test_count_path_outside_its_range_and_sgd_and_no_reg():
    from recbole_cdr_amd.fused import FusedBPRStep
    nu, ni, D = 300000, 100000, 64
    torch.manual_seed(1)
    U0, I0 = torch.randn(nu, D, device=DEV) * 0.1, torch.randn(ni, D, device=DEV) * 0.1
    for B in (2048, 20000):                                     # 2,048: below the range -> the sorted path under id_path='count' too
        a = FusedBPRStep(U0.clone(), I0.clone(), B, opt='sgd', lr=0.05, reg_weight=0.0, id_path='sort')
        b = FusedBPRStep(U0.clone(), I0.clone(), B, opt='sgd', lr=0.05, reg_weight=0.0, id_path='count')
        u, p, n = (torch.randint(0, hi, (B,), device=DEV) for hi in (nu, ni, ni))
        a.step(u, p, n); b.step(u, p, n)
        assert torch.equal(a.U, b.U) and torch.equal(a.I, b.I) and torch.equal(a.out6[:9], b.out6[:9])


def test_auto_policy_moves_a_skewed_stream_to_the_sorted_path_and_back():
    from recbole_cdr_amd.fused import FusedBPRStep
    nu, ni, D, B = 6000000, 4000000, 32, 32768
    torch.manual_seed(2)
    st = FusedBPRStep(torch.randn(nu, D, device=DEV) * 0.1, torch.randn(ni, D, device=DEV) * 0.1, B, opt='adam', lr=0.01, reg_weight=0.01)
    assert st.id_path == 'auto' and st._use_count
    mk = lambda hi: torch.randint(0, hi, (B,), device=DEV)
    for i in range(40):                                          # one hot item in a quarter of the positives
        p = mk(ni); p[: B // 4] = 5
        st.step(mk(nu), p, mk(ni)); torch.cuda.synchronize()
    assert not st._use_count
    for i in range(40):
        st.step(mk(nu), mk(ni), mk(ni)); torch.cuda.synchronize()
    assert st._use_count


def test_count_path_replays_as_a_graph():
    """The device-count form (cdr_bpr_step_fused_dev) captured with the count path inside: replays equal the eager sorted steps bit for bit."""
    nu, ni, D, B = 200000, 80000, 128, 24000
    a, b = _pair(nu, ni, D, B)
    u, p, n = (torch.randint(0, hi, (B,), device=DEV) for hi in (nu, ni, ni))
    su, sp, sn = u.clone(), p.clone(), n.clone()
    from recbole_cdr_amd import binding as B_
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        b.step(su, sp, sn)                                      # warm on the capture stream: its context's scratch and the counters are allocated here
    torch.cuda.synchronize()
    a.step(u, p, n)
    with torch.cuda.stream(s):
        with B_.capturing(g, s):
            b.step(su, sp, sn)
    torch.cuda.current_stream().wait_stream(s)
    for _ in range(3):
        u, p, n = (torch.randint(0, hi, (B,), device=DEV) for hi in (nu, ni, ni))
        su.copy_(u); sp.copy_(p); sn.copy_(n)
        g.replay(); b.replayed(1)
        a.step(u, p, n)
    torch.cuda.synchronize()
    _same(a, b)
