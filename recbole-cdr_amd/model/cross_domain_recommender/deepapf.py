"""DeepAPF on libcdrhip -- same class contract as recbole_cdr/model/cross_domain_recommender/deepapf.py:23-175.

Per domain (deepapf.py:69-152): the overlapped side has a share row and a domain-only row; both are multiplied with the other
side's row, scored by the two-layer attention MLP (ONE [2B, D] operand through the fp32-MFMA contraction: cdr_apf_prod), the pair
of scores goes through a masked softmax, the rows are merged and scored by the predict layer (cdr_apf_combine), BCE natively.
Quirks kept: the share score is masked where id > overlapped_num (strictly), i.e. the first non-overlapped id still uses its
share row; ``self.user_mlp = self.seq = ...; self.item_mlp = self.seq = ...`` -- the item MLP is ALSO registered as ``seq`` and
``named_parameters()`` reports it under that name (reference checkpoints carry user_mlp.*, seq.* and item_mlp.*)."""
import torch
import torch.nn as nn

from ... import binding as B_
from ... import functional as F_
from ...utils import InputType
from ..crossdomain_recommender import CrossDomainRecommender, xavier_normal_initialization


class DeepAPF(CrossDomainRecommender):
    input_type = InputType.POINTWISE

    def __init__(self, config, dataset):
        super().__init__(config, dataset)
        self.SOURCE_LABEL = dataset.source_domain_dataset.label_field
        self.TARGET_LABEL = dataset.target_domain_dataset.label_field
        self.mode = self.one_sided_overlap_mode()
        self.embedding_size = config['embedding_size']
        self.beta = config['beta']

        self.source_user_embedding = nn.Embedding(self.total_num_users, self.embedding_size)
        self.target_user_embedding = nn.Embedding(self.total_num_users, self.embedding_size)
        self.share_user_embedding = nn.Embedding(self.total_num_users, self.embedding_size)
        self.source_item_embedding = nn.Embedding(self.total_num_items, self.embedding_size)
        self.target_item_embedding = nn.Embedding(self.total_num_items, self.embedding_size)
        self.share_item_embedding = nn.Embedding(self.total_num_items, self.embedding_size)
        self.user_mlp = self.seq = nn.Sequential(nn.Linear(self.embedding_size, self.embedding_size), nn.ReLU(),
                                                 nn.Linear(self.embedding_size, 1, bias=False))
        self.item_mlp = self.seq = nn.Sequential(nn.Linear(self.embedding_size, self.embedding_size), nn.ReLU(),
                                                 nn.Linear(self.embedding_size, 1, bias=False))
        self.predict_layer = nn.Linear(self.embedding_size, 1, bias=False)
        self.apply(xavier_normal_initialization)

    def _forward(self, user, item, domain):
        if self.mode == 'overlap_users':
            share = F_.gather_rows(self.share_user_embedding.weight, user)
            only = F_.gather_rows(getattr(self, f'{domain}_user_embedding').weight, user)
            other = F_.gather_rows(getattr(self, f'{domain}_item_embedding').weight, item)
            key, n_over, mlp = user, self.overlapped_num_users, self.user_mlp
        else:
            other = F_.gather_rows(getattr(self, f'{domain}_user_embedding').weight, user)
            share = F_.gather_rows(self.share_item_embedding.weight, item)
            only = F_.gather_rows(getattr(self, f'{domain}_item_embedding').weight, item)
            key, n_over, mlp = item, self.overlapped_num_items, self.item_mlp
        x = F_.ApfProduct.apply(share, only, other)                                       # [2B, D]
        a = F_.linear(F_.linear(x, mlp[0].weight, mlp[0].bias, B_.ACT_RELU), mlp[2].weight, None, B_.ACT_NONE)   # [2B, 1]
        return F_.ApfCombine.apply(a, share, only, other, self.predict_layer.weight, key, int(n_over))

    def source_forward(self, user, item):
        return self._forward(user, item, 'source')

    def target_forward(self, user, item):
        return self._forward(user, item, 'target')

    def forward(self):
        pass

    @torch.no_grad()
    def predict(self, interaction):
        return self.target_forward(interaction[self.TARGET_USER_ID], interaction[self.TARGET_ITEM_ID])

    def graph_key(self):
        return ('DeepAPF',)

    def calculate_loss(self, interaction):
        p_source = self.source_forward(interaction[self.SOURCE_USER_ID], interaction[self.SOURCE_ITEM_ID])
        p_target = self.target_forward(interaction[self.TARGET_USER_ID], interaction[self.TARGET_ITEM_ID])
        return F_.BCEProbLoss.apply(p_source, interaction[self.SOURCE_LABEL]) + \
            F_.BCEProbLoss.apply(p_target, interaction[self.TARGET_LABEL])
