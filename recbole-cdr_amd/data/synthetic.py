"""Synthetic cross-domain datasets in the reference's id-space contract (dataset.py:384-399): what bench.py and the
examples feed the models with when no interaction files are available (the reference's own sample .inter files are
absent from its repository: SURVEY F3)."""
import numpy as np
import scipy.sparse as sp
import torch


class _Domain:
    def __init__(self, name, n_user, n_item, inter_feat):
        self.uid_field, self.iid_field, self.label_field = f'{name}_user_id', f'{name}_item_id', f'{name}_label'
        self._n = {self.uid_field: n_user, self.iid_field: n_item}
        self.inter_feat = inter_feat

    def num(self, field):
        return self._n[field]


class SyntheticCrossDomainDataset:
    """ids: [0]=PAD, [1,OU) overlap, [OU,OU+TO) target-only, [OU+TO,total) source-only; uniform random interactions."""

    def __init__(self, OU, TOU, SOU, OI, TOI, SOI, n_source_inter, n_target_inter, seed=2022):
        rng = np.random.RandomState(seed)
        self.num_overlap_user, self.num_overlap_item = OU, OI
        self.num_target_only_user, self.num_source_only_user = TOU, SOU
        self.num_target_only_item, self.num_source_only_item = TOI, SOI
        self.num_total_user, self.num_total_item = OU + TOU + SOU, OI + TOI + SOI
        self.overlap_id_field = 'overlap'
        self.src_users = np.concatenate([np.arange(1, OU), np.arange(OU + TOU, self.num_total_user)])
        self.src_items = np.concatenate([np.arange(1, OI), np.arange(OI + TOI, self.num_total_item)])
        self.tgt_users, self.tgt_items = np.arange(1, OU + TOU), np.arange(1, OI + TOI)
        self.s_pairs = np.unique(np.stack([rng.choice(self.src_users, n_source_inter), rng.choice(self.src_items, n_source_inter)], 1), axis=0)
        self.t_pairs = np.unique(np.stack([rng.choice(self.tgt_users, n_target_inter), rng.choice(self.tgt_items, n_target_inter)], 1), axis=0)
        s_feat = {'source_user_id': torch.from_numpy(self.s_pairs[:, 0].copy()), 'source_item_id': torch.from_numpy(self.s_pairs[:, 1].copy())}
        t_feat = {'target_user_id': torch.from_numpy(self.t_pairs[:, 0].copy()), 'target_item_id': torch.from_numpy(self.t_pairs[:, 1].copy())}
        self.source_domain_dataset = _Domain('source', OU + SOU, OI + SOI, s_feat)
        self.target_domain_dataset = _Domain('target', OU + TOU, OI + TOI, t_feat)

    def inter_matrix(self, form='coo', value_field=None, domain='source'):
        p = self.s_pairs if domain == 'source' else self.t_pairs
        return sp.coo_matrix((np.ones(len(p), dtype=np.float32), (p[:, 0], p[:, 1])),
                             shape=(self.num_total_user, self.num_total_item))

    def pointwise_batch(self, domain, S, k, rng, device):
        """recbole POINTWISE layout: S positives repeated (1+k) times, negatives k-major, labels [1]*S + [0]*(S k)."""
        pairs = self.s_pairs if domain == 'source' else self.t_pairs
        items = self.src_items if domain == 'source' else self.tgt_items
        sel = pairs[rng.randint(0, len(pairs), S)]
        u = np.tile(sel[:, 0], 1 + k)
        i = np.concatenate([sel[:, 1], rng.choice(items, S * k)])
        y = np.concatenate([np.ones(S), np.zeros(S * k)]).astype(np.float32)
        return {f'{domain}_user_id': torch.from_numpy(u).to(device), f'{domain}_item_id': torch.from_numpy(i).to(device),
                f'{domain}_label': torch.from_numpy(y).to(device)}

    def pairwise_batch(self, domain, S, k, rng, device):
        """recbole PAIRWISE layout: S positives repeated k times, ``neg_<iid>`` k-major."""
        pairs = self.s_pairs if domain == 'source' else self.t_pairs
        items = self.src_items if domain == 'source' else self.tgt_items
        sel = pairs[rng.randint(0, len(pairs), S)]
        return {f'{domain}_user_id': torch.from_numpy(np.tile(sel[:, 0], k)).to(device),
                f'{domain}_item_id': torch.from_numpy(np.tile(sel[:, 1], k)).to(device),
                f'neg_{domain}_item_id': torch.from_numpy(rng.choice(items, S * k)).to(device)}


class DeviceSyntheticDataset:
    """The same id-space contract with the interactions generated ON the device (torch): what the C5-sized end-to-end leg of bench.py
    trains on -- 50 M users x 10 M items per domain leave no room for host-side np.unique over tens of millions of pairs.
    ``s_pairs`` / ``t_pairs``: int64 [n, 2] device tensors of distinct (user, item) pairs in random order."""

    def __init__(self, OU, TOU, SOU, OI, TOI, SOI, n_source_inter, n_target_inter, device, seed=2022):
        g = torch.Generator(device=device); g.manual_seed(seed)
        self.num_overlap_user, self.num_overlap_item = OU, OI
        self.num_target_only_user, self.num_source_only_user = TOU, SOU
        self.num_target_only_item, self.num_source_only_item = TOI, SOI
        self.num_total_user, self.num_total_item = OU + TOU + SOU, OI + TOI + SOI
        self.overlap_id_field = 'overlap'

        def draw(n, lo_a, hi_a, lo_b, hi_b):
            """n ids uniform over [lo_a, hi_a) U [lo_b, hi_b)"""
            na, nb = max(hi_a - lo_a, 0), max(hi_b - lo_b, 0)
            c = torch.randint(0, na + nb, (n,), device=device, generator=g)
            return torch.where(c < na, c + lo_a, c - na + lo_b)

        def pairs(n, urange, irange):
            u, i = draw(n, *urange), draw(n, *irange)
            key = torch.unique(u * self.num_total_item + i)
            key = key[torch.randperm(key.numel(), device=device, generator=g)]
            return torch.stack([key // self.num_total_item, key % self.num_total_item], 1).contiguous()
        self.s_pairs = pairs(n_source_inter, (1, OU, OU + TOU, self.num_total_user), (1, OI, OI + TOI, self.num_total_item))
        self.t_pairs = pairs(n_target_inter, (1, OU + TOU, 0, 0), (1, OI + TOI, 0, 0))
        self.source_domain_dataset = _Domain('source', OU + SOU, OI + SOI, {'source_user_id': self.s_pairs[:, 0], 'source_item_id': self.s_pairs[:, 1]})
        self.target_domain_dataset = _Domain('target', OU + TOU, OI + TOI, {'target_user_id': self.t_pairs[:, 0], 'target_item_id': self.t_pairs[:, 1]})
