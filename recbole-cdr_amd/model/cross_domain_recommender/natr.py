"""NATR on libcdrhip -- same class contract as recbole_cdr/model/cross_domain_recommender/natr.py:23-191.

Phase 1 (SOURCE, natr.py:98-110): sigmoid-dot BCE on the source tables -- the fused gather-dot-loss kernel.
Phase 2 (TARGET, natr.py:112-168): the history of the row's user (overlap_items) or item (overlap_users) in the target domain is
looked up in the SOURCE table, pushed through the transfer layer on the fp32-MFMA contraction ([B * n_hist, Ds] x [Ds, Dt]) and
one kernel (cdr_natr_att_fwd / _bwd, one wave per batch row) does unit-level attention, masked softmax, the weighted history
sum, the domain-level gate and the score; BCE + reg_weight * RegLoss (sum of five 2-norms) natively.  The history matrices are
built on the device (data/history.py).  set_phase('TARGET') freezes the source tables as the reference does (natr.py:69-73)."""
import torch
import torch.nn as nn

from ... import binding as B_
from ... import functional as F_
from ...utils import InputType
from ..crossdomain_recommender import CrossDomainRecommender, xavier_normal_initialization


class NATR(CrossDomainRecommender):
    input_type = InputType.POINTWISE

    def __init__(self, config, dataset):
        super().__init__(config, dataset)
        self.mode = self.one_sided_overlap_mode()
        self.phase = None
        self.source_embedding_size = config['source_embedding_size']
        self.target_embedding_size = config['target_embedding_size']
        self.reg_weight = config['reg_weight']
        self.max_inter_length = config['max_inter_length']
        self.SOURCE_LABEL = dataset.source_domain_dataset.label_field
        self.TARGET_LABEL = dataset.target_domain_dataset.label_field
        if self.mode == 'overlap_users':
            self.history_user_matrix, self.history_lens, self.mask_mat = self.get_history_user_info(dataset)
        if self.mode == 'overlap_items':
            self.history_item_matrix, self.history_lens, self.mask_mat = self.get_history_item_info(dataset)

        self.source_user_embedding = nn.Embedding(self.total_num_users, self.source_embedding_size)
        self.source_item_embedding = nn.Embedding(self.total_num_items, self.source_embedding_size)
        self.target_user_embedding = nn.Embedding(self.total_num_users, self.target_embedding_size)
        self.target_item_embedding = nn.Embedding(self.total_num_items, self.target_embedding_size)
        # (the reference zero-fills the rows a domain never sees, natr.py:61-65, then re-initialises every table, :73)
        self.transfer_layer = nn.Linear(self.source_embedding_size, self.target_embedding_size)
        self.unit_attention_layer = nn.Linear(self.target_embedding_size, 1)
        self.domain_attention_layer = nn.Linear(self.target_embedding_size, 1)
        self.apply(xavier_normal_initialization)

    def set_phase(self, phase):
        self.phase = phase
        if phase == 'TARGET':
            self.source_item_embedding.weight.requires_grad = False
            self.source_user_embedding.weight.requires_grad = False

    def _history_info(self, triple):
        matrix, _, lens = triple
        matrix = matrix[:, :self.max_inter_length].contiguous().to(self.device)
        lens = lens.to(self.device)
        mask = (torch.arange(matrix.shape[1], device=matrix.device) < lens.unsqueeze(1)).float()
        return matrix, lens, mask

    def get_history_item_info(self, dataset):
        return self._history_info(dataset.history_item_matrix(domain='target'))

    def get_history_user_info(self, dataset):
        return self._history_info(dataset.history_user_matrix(domain='target'))

    def _phase1(self, user, item, label):
        return F_.PointGatherLoss.apply(B_.CDR_LOSS_BCE, self.source_user_embedding.weight, self.source_item_embedding.weight,
                                        None, None, user, item, label, 0.0)

    def phase1_forward(self, user, item):
        with torch.no_grad():
            return self._phase1(user, item, torch.zeros(user.numel(), device=user.device))[1]

    def calculate_phase1_loss(self, interaction):
        return self._phase1(interaction[self.SOURCE_USER_ID], interaction[self.SOURCE_ITEM_ID],
                            interaction[self.SOURCE_LABEL])[0].reshape(())

    def phase2_forward(self, user, item):
        user_e = F_.gather_rows(self.target_user_embedding.weight, user)
        item_e = F_.gather_rows(self.target_item_embedding.weight, item)
        if self.mode == 'overlap_items':
            key, hist, src, pu, qi = user, self.history_item_matrix, self.source_item_embedding.weight, user_e, item_e
        else:
            key, hist, src, pu, qi = item, self.history_user_matrix, self.source_user_embedding.weight, item_e, user_e
        key = key.reshape(-1)
        he = F_.gather_rows(src, hist[key])                                              # [B, n_hist, Ds]
        he = F_.linear(he, self.transfer_layer.weight, self.transfer_layer.bias, B_.ACT_NONE)   # [B, n_hist, Dt]
        return F_.NatrAttention.apply(he, pu, qi, self.mask_mat[key], self.unit_attention_layer.weight,
                                      self.unit_attention_layer.bias, self.domain_attention_layer.weight,
                                      self.domain_attention_layer.bias)

    def calculate_phase2_loss(self, interaction):
        score = self.phase2_forward(interaction[self.TARGET_USER_ID], interaction[self.TARGET_ITEM_ID])
        rec_loss = F_.BCEProbLoss.apply(score, interaction[self.TARGET_LABEL])
        reg_loss = None                                                                  # recbole RegLoss: sum of 2-norms
        for w in (self.target_user_embedding.weight, self.target_item_embedding.weight, self.transfer_layer.weight,
                  self.unit_attention_layer.weight, self.domain_attention_layer.weight):
            n = F_.FrobeniusNorm.apply(w)
            reg_loss = n if reg_loss is None else reg_loss + n
        return rec_loss + self.reg_weight * reg_loss

    def graph_key(self):
        return ('NATR', self.phase)

    def calculate_loss(self, interaction):
        if self.phase == 'SOURCE':
            return self.calculate_phase1_loss(interaction)
        elif self.phase == 'TARGET':
            return self.calculate_phase2_loss(interaction)
        return None

    @torch.no_grad()
    def predict(self, interaction):
        if self.phase == 'SOURCE':
            return self.phase1_forward(interaction[self.SOURCE_USER_ID], interaction[self.SOURCE_ITEM_ID])
        return self.phase2_forward(interaction[self.TARGET_USER_ID], interaction[self.TARGET_ITEM_ID])
