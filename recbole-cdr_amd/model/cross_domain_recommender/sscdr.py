"""SSCDR on libcdrhip -- same class contract as recbole_cdr/model/cross_domain_recommender/sscdr.py:23-259.

Metric-learning losses = row gather -> squared-norm "normalize" -> triplet margin, all native kernels with native
backward; the mapping is recbole's ``MLPLayers(..., 'tanh')`` (Linear + Tanh after EVERY layer, the last included --
SURVEY App. A) on the fp32 MFMA contraction; scoring is the -||u - i||^2 epilogue of the same contraction.
The semi-supervised (interacted, non-interacted) ids are drawn on the host from numpy's global RNG exactly as the
reference does inside its loss (sscdr.py:89-118) so that a seeded run samples the same ids -- or, with
``config['sscdr_device_sampler'] = True``, by a kernel over the device-resident interaction lists (``sample_device``: same
distribution and constraints, counter-based RNG, no host work inside the loss, so the OVERLAP step replays as a hipGraph).
"""
import numpy as np
import torch
import torch.nn as nn

from ... import binding as B_
from ... import functional as F_
from ...utils import InputType
from ..crossdomain_recommender import CrossDomainRecommender, xavier_normal_initialization


class MLPLayers(nn.Module):
    """recbole.model.layers.MLPLayers restricted to what SSCDR uses (dropout=0, bn=False): parameter names match
    (``mlp_layers.<3i+1>.weight``) so reference checkpoints load."""

    def __init__(self, layers, activation='tanh'):
        super().__init__()
        mods = []
        for d_in, d_out in zip(layers[:-1], layers[1:]):
            mods += [nn.Dropout(p=0.0), nn.Linear(d_in, d_out), nn.Tanh()]
        self.mlp_layers = nn.Sequential(*mods)

    def forward(self, x):
        for m in self.mlp_layers:
            if isinstance(m, nn.Linear):
                x = F_.linear(x, m.weight, m.bias, B_.ACT_TANH)
        return x


class SSCDR(CrossDomainRecommender):
    input_type = InputType.PAIRWISE

    def __init__(self, config, dataset):
        super().__init__(config, dataset)
        self.mode = self.one_sided_overlap_mode()
        self.phase = None
        self.embedding_size = config['embedding_size']
        self.lamda = config['lambda']
        self.margin = config['margin']
        self.mlp_hidden_size = list(config['mlp_hidden_size'])
        self.device_sampler = bool(config['sscdr_device_sampler']) if 'sscdr_device_sampler' in config else False
        self.fused_map = bool(config['sscdr_fused_map']) if 'sscdr_fused_map' in config else True    # (with the device sampler only)
        self.sampler_seed = int(config['seed']) if 'seed' in config else 2022
        self.mapping_layer = MLPLayers([self.embedding_size] + self.mlp_hidden_size + [self.embedding_size])
        if self.mode == 'overlap_users':
            self.user_interacted_items = self.build_interacted_items(dataset, mode='user')
        elif self.mode == 'overlap_items':
            self.item_interacted_users = self.build_interacted_items(dataset, mode='item')
        self.source_user_embedding = nn.Embedding(self.total_num_users, self.embedding_size)
        self.source_item_embedding = nn.Embedding(self.total_num_items, self.embedding_size)
        self.target_user_embedding = nn.Embedding(self.total_num_users, self.embedding_size)
        self.target_item_embedding = nn.Embedding(self.total_num_items, self.embedding_size)
        self.apply(xavier_normal_initialization)

    def build_interacted_items(self, dataset, mode='user'):
        ds = dataset.source_domain_dataset
        uids = ds.inter_feat[ds.uid_field].cpu().numpy()
        iids = ds.inter_feat[ds.iid_field].cpu().numpy()
        if mode == 'user':
            out = [[] for _ in range(self.total_num_users)]
            for uid, iid in zip(uids, iids):
                out[uid].append(iid)
        else:
            out = [[] for _ in range(self.total_num_items)]
            for iid, uid in zip(iids, uids):
                out[iid].append(uid)
        return out

    def sample(self, ids, mode='user'):
        """Host-side, numpy global RNG, same draw order per id as sscdr.py:89-118 (and the same cache mutation).  The reference calls
        ``np.random.choice(<python list>, size=1)``: a list -> array conversion of the whole candidate list per draw (0.33 ms for 7 k
        candidates, 10 ms for a batch of 100 ids).  Legacy ``RandomState.choice`` without ``p`` draws ``randint(0, len(a), size)`` and
        indexes, so ``randint`` on a cached array consumes the global stream identically (golden-pinned: ``aux/sampled_pos|neg``)."""
        ids = ids.cpu().numpy()
        interacted = np.zeros_like(ids)
        non_interacted = np.zeros_like(ids)
        cache = self.__dict__.setdefault('_cand_cache', {})
        if mode not in cache:
            if mode == 'user':
                cand = list(range(self.overlapped_num_items)) + list(range(self.target_num_items, self.total_num_items))
            else:
                cand = list(range(self.overlapped_num_users)) + list(range(self.target_num_users, self.total_num_users))
            cache[mode] = np.asarray(cand)
        cand = cache[mode]
        lists = self.user_interacted_items if mode == 'user' else self.item_interacted_users
        n_cand = len(cand)
        for index, id_ in enumerate(ids):
            h = lists[id_]
            if len(h) == 0:
                h.append(0)
            c = cand[np.random.randint(0, n_cand, size=1)[0]]
            while c in h:
                c = cand[np.random.randint(0, n_cand, size=1)[0]]
            interacted[index] = h[np.random.randint(0, len(h), size=1)[0]]
            non_interacted[index] = c
        return torch.from_numpy(interacted).to(self.device), torch.from_numpy(non_interacted).to(self.device)

    def _device_lists(self, mode):
        """The interaction lists of ``build_interacted_items`` as a device CSR (entries ascending per id, repeats kept), the candidate
        ranges of sscdr.py:94-95,106-107 and the sampler's device state; built once, from the lists the constructor made."""
        cache = self.__dict__.setdefault('_dev_lists', {})
        if mode not in cache:
            lists = self.user_interacted_items if mode == 'user' else self.item_interacted_users
            lens = np.fromiter((len(h) for h in lists), dtype=np.int64, count=len(lists))
            flat = np.fromiter((x for h in lists for x in sorted(h)), dtype=np.int64, count=int(lens.sum()))
            indptr = np.zeros(len(lists) + 1, dtype=np.int64)
            np.cumsum(lens, out=indptr[1:])
            dev = self.source_user_embedding.weight.device
            if mode == 'user':
                rng = (0, self.overlapped_num_items, self.target_num_items, self.total_num_items)
            else:
                rng = (0, self.overlapped_num_users, self.target_num_users, self.total_num_users)
            cache[mode] = (torch.from_numpy(indptr).to(dev), torch.from_numpy(flat if flat.size else np.zeros(1, dtype=np.int64)).to(dev), rng,
                           torch.zeros(1, device=dev, dtype=torch.int64), torch.zeros(1, device=dev, dtype=torch.int32))
        return cache[mode]

    def sample_device(self, ids, mode='user'):
        """``sample`` on the device (csrc/cdr_sampler.hip: sscdr_pair_sample_kernel): same distribution -- interacted uniform over the
        id's interaction list (an empty list counts as [0]), non-interacted uniform over the candidates not in it -- but its own
        counter-based RNG stream (``config['seed']``, a device call counter), and the cached lists are not mutated."""
        indptr, indices, (lo0, hi0, lo1, hi1), calls, fail = self._device_lists(mode)
        ids = ids.reshape(-1).contiguous().to(torch.int64)
        n = ids.numel()
        pos = torch.empty(n, device=ids.device, dtype=torch.int64)
        neg = torch.empty(n, device=ids.device, dtype=torch.int64)
        B_.call('cdr_sscdr_pair_sample', B_.stream(), B_.i64(ids), n, lo0, hi0, lo1, hi1, B_.i64(indptr), B_.i64(indices),
                (self.sampler_seed * 0x9E3779B1) & 0xFFFFFFFFFFFFFFFF, B_.i64(calls), B_.i64(pos), B_.i64(neg), B_.raw(fail))
        if not self.__dict__.get('_bump_in_gather', False):      # (the fused map loss advances the counter in its gather launch)
            B_.call('cdr_inc_i64', B_.stream(), B_.i64(calls))
        return pos, neg

    embedding_normalize = staticmethod(F_.sqnorm_normalize)

    def set_phase(self, phase):
        self.phase = phase

    def _domain_loss(self, interaction, domain):
        U = getattr(self, f'{domain}_user_embedding').weight
        I = getattr(self, f'{domain}_item_embedding').weight
        pre = domain.upper()
        ue = F_.gather_rows(U, interaction[getattr(self, f'{pre}_USER_ID')])
        pe = F_.gather_rows(I, interaction[getattr(self, f'{pre}_ITEM_ID')])
        ne = F_.gather_rows(I, interaction[getattr(self, f'{pre}_NEG_ITEM_ID')])
        return F_.TripletMarginLoss.apply(F_.sqnorm_normalize(ue), F_.sqnorm_normalize(pe), F_.sqnorm_normalize(ne),
                                          self.margin)

    def calculate_source_loss(self, interaction):
        return self._domain_loss(interaction, 'source')

    def calculate_target_loss(self, interaction):
        return self._domain_loss(interaction, 'target')

    def calculate_map_loss(self, interaction):
        idx = interaction[self.OVERLAP_ID].squeeze(1)
        a, b = ('user', 'item') if self.mode == 'overlap_users' else ('item', 'user')
        if self.device_sampler and self.fused_map:
            # the same loss in 7 launches instead of 21 (and 11 instead of 45 backward): sampler, ONE gather of the four row sets, the
            # mapping once on the stacked [source ; interacted ; non-interacted] rows, ONE loss kernel that also leaves the gradients
            self.__dict__['_bump_in_gather'] = True
            try:
                pos, neg = self.sample_device(idx, mode=a)
            finally:
                self.__dict__['_bump_in_gather'] = False
            X3, tgt = F_.GatherMapRows.apply(getattr(self, f'source_{a}_embedding').weight, getattr(self, f'target_{a}_embedding').weight,
                                             getattr(self, f'source_{b}_embedding').weight, idx, pos, neg, self._device_lists(a)[3])
            total, _ = F_.SSCDRMapLoss.apply(self.mapping_layer(X3), tgt, self.margin, self.lamda)
            return total
        src = F_.gather_rows(getattr(self, f'source_{a}_embedding').weight, idx)
        tgt = F_.gather_rows(getattr(self, f'target_{a}_embedding').weight, idx)
        loss_s = F_.mse_loss(self.mapping_layer(src), tgt)
        pos, neg = self.sample_device(idx, mode=a) if self.device_sampler else self.sample(idx, mode=a)
        other = getattr(self, f'source_{b}_embedding').weight
        mp = self.mapping_layer(F_.gather_rows(other, pos))
        mn = self.mapping_layer(F_.gather_rows(other, neg))
        loss_u = F_.TripletMarginLoss.apply(F_.sqnorm_normalize(tgt), F_.sqnorm_normalize(mp), F_.sqnorm_normalize(mn),
                                            self.margin)
        return loss_s + self.lamda * loss_u

    def graph_key(self):
        # OVERLAP with the reference's numpy sampler: host draws inside the loss -- not capturable; the device sampler is
        if self.phase == 'OVERLAP' and not self.device_sampler:
            return None
        return ('SSCDR', self.phase, self.device_sampler and self.fused_map)

    def graph_state_tensors(self):
        """Device-side call counters of the in-loss sampler (a captured step's warm-up must put them back)."""
        return [v[3] for v in self.__dict__.get('_dev_lists', {}).values()]

    def calculate_loss(self, interaction):
        if self.phase == 'SOURCE':
            return self.calculate_source_loss(interaction)
        elif self.phase == 'OVERLAP':
            return self.calculate_map_loss(interaction)
        else:
            return self.calculate_target_loss(interaction)

    # ---- scoring --------------------------------------------------------------------------------------------------
    def _mapped_rows(self, kind, ids, n_overlap):
        src = F_.gather_rows(getattr(self, f'source_{kind}_embedding').weight, ids)
        return F_.select_mapped(self.mapping_layer(src), getattr(self, f'target_{kind}_embedding').weight, ids, n_overlap)

    @staticmethod
    def _neg_rowdist(a, b):
        """-sum((a-b)^2, 1) for explicit pairs: the triplet kernel's distance with eps=0, squared."""
        rows, D = a.shape
        out = torch.empty(1, device=a.device, dtype=torch.float32)
        dap = torch.empty(rows, device=a.device, dtype=torch.float32)
        dan = torch.empty(rows, device=a.device, dtype=torch.float32)
        a_, b_ = a.contiguous(), b.contiguous()
        B_.call('cdr_triplet_fwd', B_.ctx(a.device), B_.stream(), B_.f32(a_), B_.f32(b_), B_.f32(b_), rows, D, 0.0, 0.0,
                B_.f32(out), B_.f32(dap), B_.f32(dan))
        # d^2 with the sign flipped: one more native elementwise pass (mse-style) would only restate dap*dap
        return -(dap * dap)

    @torch.no_grad()
    def predict(self, interaction):
        if self.phase in ('SOURCE', 'TARGET'):
            d = self.phase.lower()
            ue = F_.sqnorm_normalize(F_.gather_rows(getattr(self, f'{d}_user_embedding').weight,
                                                    interaction[getattr(self, f'{self.phase}_USER_ID')]))
            ie = F_.sqnorm_normalize(F_.gather_rows(getattr(self, f'{d}_item_embedding').weight,
                                                    interaction[getattr(self, f'{self.phase}_ITEM_ID')]))
            return self._neg_rowdist(ue, ie)
        user, item = interaction[self.TARGET_USER_ID], interaction[self.TARGET_ITEM_ID]
        if self.mode == 'overlap_users':
            ue = self._mapped_rows('user', user, self.overlapped_num_users)
            ie = F_.gather_rows(self.target_item_embedding.weight, item)
        else:
            ue = F_.gather_rows(self.target_user_embedding.weight, user)
            ie = self._mapped_rows('item', item, self.overlapped_num_items)
        return self._neg_rowdist(F_.sqnorm_normalize(ue), F_.sqnorm_normalize(ie))

    @torch.no_grad()
    def full_sort_predict(self, interaction):
        OI, TI = self.overlapped_num_items, self.target_num_items
        if self.phase == 'SOURCE':
            ue = F_.sqnorm_normalize(F_.gather_rows(self.source_user_embedding.weight, interaction[self.SOURCE_USER_ID]))
            W = self.source_item_embedding.weight
            all_item = torch.cat([F_.sqnorm_normalize(W[:OI]), F_.sqnorm_normalize(W[TI:])], dim=0)
        elif self.phase == 'TARGET':
            ue = F_.sqnorm_normalize(F_.gather_rows(self.target_user_embedding.weight, interaction[self.TARGET_USER_ID]))
            all_item = F_.sqnorm_normalize(self.target_item_embedding.weight[:TI])
        else:
            user = interaction[self.TARGET_USER_ID]
            if self.mode == 'overlap_users':
                ue = self._mapped_rows('user', user, self.overlapped_num_users)
                all_item = self.target_item_embedding.weight[:TI]
            else:
                ue = F_.gather_rows(self.target_user_embedding.weight, user)
                ov = self.mapping_layer(self.source_item_embedding.weight[:OI])
                all_item = torch.cat([ov, self.target_item_embedding.weight[OI:TI]], dim=0)
            ue = F_.sqnorm_normalize(ue)
            all_item = F_.sqnorm_normalize(all_item)
        return F_.fullsort_neg_sqdist(ue, all_item).view(-1)
