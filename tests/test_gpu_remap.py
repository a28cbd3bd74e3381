"""The overlap id remap on the device (csrc/cdr_remap_dev.hip; dataset.py:344-445 + :109-123): bit-exact with the reference's
recorded dictionaries (the six golden remap cases), with the oracle restatement on random tokens of awkward shapes (prefixes, empty
tokens, NUL bytes, multi-chunk lengths, NaN, '[PAD]' in every class), and with the host form at 2 M tokens."""
import numpy as np
import pytest
import torch

import recbole_cdr_amd  # noqa: F401
from recbole_cdr_amd.data import overlap_remap, overlap_remap_packed
from golden_util import Golden, cases
from helpers import DEV

pytestmark = pytest.mark.gpu


def _tokens(g, key):
    toks = [str(t) for t in g[f'in/{key}_tokens']]
    nan = g[f'in/{key}_isnan'] if g.has(f'in/{key}_isnan') else np.zeros(len(toks), bool)
    return [None if m else t for t, m in zip(toks, nan)]


@pytest.mark.parametrize('name', cases('remap_'))
def test_device_remap_golden(name):
    g = Golden(name)
    su, si, tu, ti = (_tokens(g, k) for k in ('source_user', 'source_item', 'target_user', 'target_item'))
    sfeat = [str(t) for t in g['in/source_user_feat_tokens']] if g.has('in/source_user_feat_tokens') else []
    tfeat = [str(t) for t in g['in/target_user_feat_tokens']] if g.has('in/target_user_feat_tokens') else []
    ru = overlap_remap(su + sfeat, tu + tfeat, device=DEV)
    ri = overlap_remap(si, ti, device=DEV)
    np.testing.assert_array_equal(ru.source_ids[:len(su)], g['applied/source_user'])
    np.testing.assert_array_equal(ru.target_ids[:len(tu)], g['applied/target_user'])
    np.testing.assert_array_equal(ri.source_ids, g['applied/source_item'])
    np.testing.assert_array_equal(ri.target_ids, g['applied/target_item'])
    assert (ru.num_overlap, ru.num_source_only, ru.num_target_only, ru.num_total) == tuple(
        int(g[f'count/{k}']) for k in ('num_overlap_user', 'num_source_only_user', 'num_target_only_user', 'num_total_user'))
    assert (ri.num_overlap, ri.num_source_only, ri.num_target_only, ri.num_total) == tuple(
        int(g[f'count/{k}']) for k in ('num_overlap_item', 'num_source_only_item', 'num_target_only_item', 'num_total_item'))
    host_u, host_i = overlap_remap(su + sfeat, tu + tfeat), overlap_remap(si, ti)
    for a, b in ((ru, host_u), (ri, host_i)):
        np.testing.assert_array_equal(a.source_ids, b.source_ids); np.testing.assert_array_equal(a.target_ids, b.target_ids)


def _awkward(rng, n, lo, hi):
    """Tokens that stress byte order: decimal ids of mixed length ('u10' < 'u2'), shared prefixes, tokens longer than one and two 8-byte
    chunks, the empty token, non-ASCII UTF-8, '[PAD]', and NaN (None)."""
    out = []
    for v in rng.randint(lo, hi, n):
        kind = v % 11
        if kind == 0:
            out.append('u%d' % v)
        elif kind == 1:
            out.append('item-with-a-long-common-prefix-%07d' % v)
        elif kind == 2:
            out.append('x' * (v % 19))                       # prefixes of one another, including ''
        elif kind == 3:
            out.append('é中%d' % (v % 97))          # multi-byte UTF-8
        elif kind == 4:
            out.append(None if v % 5 == 0 else 'ab')
        elif kind == 5:
            out.append('[PAD]' if v % 3 == 0 else '[PAD]x')
        elif kind == 6:
            out.append('12345678' + str(v % 13))             # differs only behind the first chunk
        else:
            out.append(str(v))
    return out


@pytest.mark.parametrize('case', ['pad_overlap', 'pad_source_only', 'pad_target_only', 'disjoint', 'empty_source', 'all_nan_target'])
def test_device_remap_vs_oracle_awkward_tokens(case):
    from oracle import remap as oremap
    rng = np.random.RandomState(7)
    s, t = _awkward(rng, 30000, 0, 40000), _awkward(rng, 25000, 20000, 70000)
    strip = lambda xs: [x for x in xs if x != '[PAD]']
    if case == 'pad_source_only':
        t = strip(t); s = s + ['[PAD]']
    elif case == 'pad_target_only':
        s = strip(s); t = t + ['[PAD]']
    elif case == 'pad_overlap':
        s, t = s + ['[PAD]'], t + ['[PAD]']
    elif case == 'disjoint':
        s, t = ['s' + x for x in strip(s) if x is not None], ['t' + x for x in strip(t) if x is not None]
    elif case == 'empty_source':
        s = []
    elif case == 'all_nan_target':
        t = [None] * 100
    r = overlap_remap(s, t, device=DEV)
    ms, _, mt, _, counts = oremap.overlap_remap(s, ['x'], t, ['y'])
    np.testing.assert_array_equal(r.source_ids, oremap.apply_remap(s, ms) if s else np.zeros(0, np.int64))
    np.testing.assert_array_equal(r.target_ids, oremap.apply_remap(t, mt))
    assert (r.num_overlap, r.num_source_only, r.num_target_only, r.num_total) == (
        counts['num_overlap_user'], counts['num_source_only_user'], counts['num_target_only_user'], counts['num_total_user'])
    h = overlap_remap(s, t)                                   # the host form agrees too
    np.testing.assert_array_equal(r.source_ids, h.source_ids); np.testing.assert_array_equal(r.target_ids, h.target_ids)


def test_device_remap_nul_bytes_and_prefix_order_raw():
    """Raw byte tokens through the packed interface: a token that is a prefix of another modulo trailing NUL bytes must sort first
    (zero padding of the 8-byte chunks + the length as the least significant key)."""
    toks = [b'ab', b'ab\x00', b'ab\x00\x00', b'ab\x00c', b'', b'\x00', b'abcdefgh', b'abcdefgh\x00', b'abcdefghi', b'\xff' * 9, b'\xff' * 8]
    order = sorted(toks)
    def pack(ts):
        off = np.zeros(len(ts) + 1, np.int64); np.cumsum([len(x) for x in ts], out=off[1:])
        return np.frombuffer(b''.join(ts), np.uint8) if off[-1] else np.zeros(0, np.uint8), off, None
    rng = np.random.RandomState(0)
    s = [toks[i] for i in rng.randint(0, len(toks), 200)] + toks
    sid, tid, counts, passes = overlap_remap_packed(pack(s), pack(toks), DEV)
    assert passes == 3 and counts.tolist() == [len(toks) + 1, 0, 0, len(toks) + 1]
    want = {tok: 1 + i for i, tok in enumerate(order)}
    assert sid.tolist() == [want[x] for x in s] and tid.tolist() == [want[x] for x in toks]


def test_device_remap_two_million_tokens_vs_host_form_and_oracle():
    """bit-exact at 2 M token occurrences (1.2 M source + 0.8 M target, ~60 % duplicates, one third overlap)
    against the single-threaded host form, and against the oracle on the first 300 k of each side."""
    from oracle import remap as oremap
    rng = np.random.RandomState(11)
    mk = lambda ids: np.char.add('u', ids.astype(str)).tolist()
    s, t = mk(rng.randint(0, 600_000, 1_200_000)), mk(rng.randint(400_000, 900_000, 800_000))
    dev, host = overlap_remap(s, t, device=DEV), overlap_remap(s, t)
    np.testing.assert_array_equal(dev.source_ids, host.source_ids); np.testing.assert_array_equal(dev.target_ids, host.target_ids)
    assert (dev.num_overlap, dev.num_source_only, dev.num_target_only, dev.num_total) == (
        host.num_overlap, host.num_source_only, host.num_target_only, host.num_total)
    s2, t2 = s[:300_000], t[:300_000]
    r = overlap_remap(s2, t2, device=DEV)
    ms, _, mt, _, counts = oremap.overlap_remap(s2, ['x'], t2, ['y'])
    np.testing.assert_array_equal(r.source_ids, oremap.apply_remap(s2, ms)); np.testing.assert_array_equal(r.target_ids, oremap.apply_remap(t2, mt))
    assert r.num_total == counts['num_total_user']
