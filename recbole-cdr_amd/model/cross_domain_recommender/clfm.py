"""CLFM on libcdrhip -- same class contract as recbole_cdr/model/cross_domain_recommender/clfm.py:23-145.

Per domain: user rows [B, Du] -> factors = user_e [shared_linear ; <domain>_only_linear]^T (ONE fp32-MFMA contraction over the stacked
weight) -> sigmoid(factors . item_row) -> BCE in the fused gather-dot-loss kernel (the factors play the "user table", addressed
by row number) + reg_weight * EmbLoss of the gathered ego rows.  Scoring: factors x item slab on the MFMA full-sort kernel.
Quirk kept: ``target_item_embedding_size`` is read from ``config['source_item_embedding_size']`` (clfm.py:39)."""
import torch
import torch.nn as nn

from ... import binding as B_
from ... import functional as F_
from ...utils import InputType
from ..crossdomain_recommender import CrossDomainRecommender, xavier_normal_initialization


class CLFM(CrossDomainRecommender):
    input_type = InputType.POINTWISE

    def __init__(self, config, dataset):
        super().__init__(config, dataset)
        self.SOURCE_LABEL = dataset.source_domain_dataset.label_field
        self.TARGET_LABEL = dataset.target_domain_dataset.label_field
        self.user_embedding_size = config['user_embedding_size']
        self.source_item_embedding_size = config['source_item_embedding_size']
        self.target_item_embedding_size = config['source_item_embedding_size']
        self.share_embedding_size = config['share_embedding_size']
        self.alpha = config['alpha']
        self.reg_weight = config['reg_weight']
        assert 0 <= self.share_embedding_size <= self.source_item_embedding_size and \
            0 <= self.share_embedding_size <= self.target_item_embedding_size

        self.source_user_embedding = nn.Embedding(self.total_num_users, self.user_embedding_size)
        self.target_user_embedding = nn.Embedding(self.total_num_users, self.user_embedding_size)
        self.source_item_embedding = nn.Embedding(self.total_num_items, self.source_item_embedding_size)
        self.target_item_embedding = nn.Embedding(self.total_num_items, self.target_item_embedding_size)
        if self.share_embedding_size > 0:
            self.shared_linear = nn.Linear(self.user_embedding_size, self.share_embedding_size, bias=False)
        if self.source_item_embedding_size - self.share_embedding_size > 0:
            self.source_only_linear = nn.Linear(self.user_embedding_size,
                                                self.source_item_embedding_size - self.share_embedding_size, bias=False)
        if self.target_item_embedding_size - self.share_embedding_size > 0:
            self.target_only_linear = nn.Linear(self.user_embedding_size,
                                                self.target_item_embedding_size - self.share_embedding_size, bias=False)
        self.apply(xavier_normal_initialization)

    def _tables(self, domain):
        return getattr(self, f'{domain}_user_embedding').weight, getattr(self, f'{domain}_item_embedding').weight

    def _factors(self, user_e, domain):
        """cat([shared_linear(u), <domain>_only_linear(u)], 1) as one contraction over the row-stacked weights (clfm.py:77-85)."""
        ws = []
        if self.share_embedding_size > 0:
            ws.append(self.shared_linear.weight)
        only = getattr(self, f'{domain}_only_linear', None)
        if only is not None:
            ws.append(only.weight)
        w = ws[0] if len(ws) == 1 else torch.cat(ws, dim=0)
        return F_.linear(user_e, w, None, B_.ACT_NONE)

    def _loss_and_prob(self, user, item, label, domain):
        U, I = self._tables(domain)
        fac = self._factors(F_.gather_rows(U, user), domain)
        rows = torch.arange(fac.shape[0], device=fac.device, dtype=torch.int64)
        return F_.PointGatherLoss.apply(B_.CDR_LOSS_BCE, fac, I, None, None, rows, item, label, 0.0)

    def _emb_loss(self, user, item, domain):
        U, I = self._tables(domain)
        if U.shape[1] == I.shape[1]:
            return F_.EmbLossRows.apply(U, I, user, item)
        # recbole EmbLoss with differently sized tables: (||U[u]||_F + ||I[i]||_F) / B
        return (F_.FrobeniusNorm.apply(F_.gather_rows(U, user)) + F_.FrobeniusNorm.apply(F_.gather_rows(I, item))) / item.numel()

    def source_forward(self, user, item):
        with torch.no_grad():
            return self._loss_and_prob(user, item, torch.zeros(user.numel(), device=user.device), 'source')[1]

    def target_forward(self, user, item):
        with torch.no_grad():
            return self._loss_and_prob(user, item, torch.zeros(user.numel(), device=user.device), 'target')[1]

    def _rows(self, n, dev):
        """arange(n) on the device, made once per batch length (the factor rows of a batch are addressed 0..n-1)."""
        cache = self.__dict__.setdefault('_row_ids', {})
        key = (n, str(dev))
        if key not in cache:
            cache[key] = torch.arange(n, device=dev, dtype=torch.int64)
        return cache[key]

    def graph_key(self):
        return ('CLFM',)

    def calculate_loss(self, interaction):
        su, si, sl = interaction[self.SOURCE_USER_ID], interaction[self.SOURCE_ITEM_ID], interaction[self.SOURCE_LABEL]
        tu, ti, tl = interaction[self.TARGET_USER_ID], interaction[self.TARGET_ITEM_ID], interaction[self.TARGET_LABEL]
        (Us, Is), (Ut, It) = self._tables('source'), self._tables('target')
        fs, ft = self._factors(F_.gather_rows(Us, su), 'source'), self._factors(F_.gather_rows(Ut, tu), 'target')
        if fs.shape[1] == ft.shape[1] and fs.shape[1] % 4 == 0:
            # both domains' gather-dot-BCE in one launch each way (functional.TwoPointLoss)
            bce_s, bce_t, _, _ = F_.TwoPointLoss.apply(B_.CDR_LOSS_BCE, fs, Is, ft, It, self._rows(fs.shape[0], fs.device), si, sl,
                                                       self._rows(ft.shape[0], ft.device), ti, tl)
        else:
            bce_s, _ = F_.PointGatherLoss.apply(B_.CDR_LOSS_BCE, fs, Is, None, None, self._rows(fs.shape[0], fs.device), si, sl, 0.0)
            bce_t, _ = F_.PointGatherLoss.apply(B_.CDR_LOSS_BCE, ft, It, None, None, self._rows(ft.shape[0], ft.device), ti, tl, 0.0)
        out = [bce_s + self.reg_weight * self._emb_loss(su, si, 'source'), bce_t + self.reg_weight * self._emb_loss(tu, ti, 'target')]
        return out[0] * self.alpha + out[1] * (1 - self.alpha)

    @torch.no_grad()
    def predict(self, interaction):
        return self.target_forward(interaction[self.TARGET_USER_ID], interaction[self.TARGET_ITEM_ID])

    @torch.no_grad()
    def full_sort_predict(self, interaction):
        user_e = F_.gather_rows(self.target_user_embedding.weight, interaction[self.TARGET_USER_ID])
        fac = self._factors(user_e, 'target')
        return F_.fullsort_scores(fac, self.target_item_embedding.weight[:self.target_num_items]).view(-1)
