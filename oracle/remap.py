"""Integer paths restated (oracle; test infrastructure only) -- bit-exact targets.

  overlap_remap     : CrossDomainDataset.calculate_user_item_from_both_domain, data/dataset.py:344-445
  apply_remap       : CrossDomainSingleDataset._remap_fields, data/dataset.py:109-123
  revoke_map        : CrossDomainFullSortEvalDataLoader._set_user_property, data/dataloader.py:240-247
  source_id_lists   : CrossDomainSourceSampler.__init__, sampler/crossdomain_sampler.py:212-217
  BothModeSchedule  : CrossDomainDataloader.__next__/_next_batch_data/__len__, data/dataloader.py:114-162

Tokens are Python ``str``; order is Python's str order (code-point order == UTF-8 byte order), so 'u10' < 'u2'.
"""
import numpy as np

PAD = '[PAD]'


def _one_field(source_tokens, target_tokens):
    s = {t for t in source_tokens if t is not None}
    t = {x for x in target_tokens if x is not None}
    overlap = sorted(s & t)
    s_only = sorted(s - t)
    t_only = sorted(t - s)
    n_ov = len(overlap) + 1                                    # PAD counted (dataset.py:384)
    ov = dict(zip(overlap, range(1, n_ov)))
    ov[PAD] = 0
    t_map = dict(zip(t_only, range(n_ov, n_ov + len(t_only))))                    # target-only first
    s_map = dict(zip(s_only, range(n_ov + len(t_only), n_ov + len(t_only) + len(s_only))))
    src = dict(ov); src.update(s_map)
    tgt = dict(ov); tgt.update(t_map)
    counts = {'num_overlap': n_ov, 'num_source_only': len(s_only), 'num_target_only': len(t_only),
              'num_total': n_ov + len(s_only) + len(t_only)}
    return src, tgt, counts


def overlap_remap(source_users, source_items, target_users, target_items):
    """``None`` marks a NaN token (dropped from the *-only lists, dataset.py:368-371).
    Returns (source_user_map, source_item_map, target_user_map, target_item_map, counts)."""
    su, tu, cu = _one_field(source_users, target_users)
    si, ti, ci = _one_field(source_items, target_items)
    counts = {f'{k}_user': v for k, v in cu.items()}
    counts.update({f'{k}_item': v for k, v in ci.items()})
    return su, si, tu, ti, counts


def apply_remap(tokens, mapping):
    return np.array([-1 if t is None else mapping.get(t, t) for t in tokens], dtype=np.int64)


def revoke_map(item_ids, overlap_item_num, target_only_item_num):
    item_ids = np.asarray(item_ids, dtype=np.int64)
    return np.where(item_ids < overlap_item_num, item_ids, item_ids - target_only_item_num)


def source_id_lists(OU, TOU, SOU, OI, TOI, SOI):
    items = np.array(list(range(1, OI)) + list(range(OI + TOI, OI + TOI + SOI)), dtype=np.int64)
    users = np.array(list(range(1, OU)) + list(range(OU + TOU, OU + TOU + SOU)), dtype=np.int64)
    return users, items


class BothModeSchedule:
    """Which (source batch, target batch, overlap batch) indices one epoch serves in each loader state.
    BOTH: epoch length = target loader; the source loader wraps WITHOUT reshuffling and both cursors reset when the
    target is exhausted (dataloader.py:119-123,156-161)."""

    def __init__(self, n_source, n_target, n_overlap):
        self.n_source, self.n_target, self.n_overlap = n_source, n_target, n_overlap

    def length(self, state):
        return {'SOURCE': self.n_source, 'TARGET': self.n_target, 'BOTH': self.n_target, 'OVERLAP': self.n_overlap}[state]

    def epoch(self, state):
        if state == 'SOURCE':
            return [(b, -1, -1) for b in range(self.n_source)]
        if state == 'TARGET':
            return [(-1, b, -1) for b in range(self.n_target)]
        if state == 'OVERLAP':
            return [(-1, -1, b) for b in range(self.n_overlap)]
        return [(b % self.n_source, b, -1) for b in range(self.n_target)]
