"""CMF on libcdrhip -- same class contract as recbole_cdr/model/cross_domain_recommender/cmf.py:23-112.
Both domains share one user and one item table; each domain's loss is one fused gather-dot-sigmoid-BCE(+EmbLoss)
launch; scoring is the fp32-MFMA contraction over item rows [0, target_num_items)."""
import torch
import torch.nn as nn

from ... import binding as B_
from ... import functional as F_
from ...utils import InputType
from ..crossdomain_recommender import CrossDomainRecommender, xavier_normal_initialization


class CMF(CrossDomainRecommender):
    input_type = InputType.POINTWISE

    def __init__(self, config, dataset):
        super().__init__(config, dataset)
        self.SOURCE_LABEL = dataset.source_domain_dataset.label_field
        self.TARGET_LABEL = dataset.target_domain_dataset.label_field
        self.embedding_size = config['embedding_size']
        self.alpha = config['alpha']
        self.lamda = config['lambda']
        self.gamma = config['gamma']
        self.user_embedding = nn.Embedding(self.total_num_users, self.embedding_size)
        self.item_embedding = nn.Embedding(self.total_num_items, self.embedding_size)
        self.apply(xavier_normal_initialization)

    def _loss_and_prob(self, user, item, label, reg):
        return F_.PointGatherLoss.apply(B_.CDR_LOSS_BCE, self.user_embedding.weight, self.item_embedding.weight,
                                        None, None, user, item, label, reg)

    def forward(self, user, item):
        zeros = torch.zeros(user.numel(), device=user.device, dtype=torch.float32)
        with torch.no_grad():
            _, p = self._loss_and_prob(user, item, zeros, 0.0)
        return p

    def graph_key(self):
        return ('CMF',)

    def calculate_loss(self, interaction):
        # both domains' batches on the shared tables as ONE autograd node (two loss launches forward, two scatter launches into one pair
        # of gradient buffers backward)
        total, _ = F_.TwoDomainPointLoss.apply(
            B_.CDR_LOSS_BCE, self.user_embedding.weight, self.item_embedding.weight,
            interaction[self.SOURCE_USER_ID], interaction[self.SOURCE_ITEM_ID], interaction[self.SOURCE_LABEL], self.lamda,
            interaction[self.TARGET_USER_ID], interaction[self.TARGET_ITEM_ID], interaction[self.TARGET_LABEL], self.gamma, self.alpha)
        return total

    @torch.no_grad()
    def predict(self, interaction):
        return self.forward(interaction[self.TARGET_USER_ID], interaction[self.TARGET_ITEM_ID])

    @torch.no_grad()
    def full_sort_predict(self, interaction):
        user_e = F_.gather_rows(self.user_embedding.weight, interaction[self.TARGET_USER_ID])
        score = F_.fullsort_scores(user_e, self.item_embedding.weight[:self.target_num_items])
        return score.view(-1)

    @torch.no_grad()
    def full_sort_topk(self, interaction, k, hist_indptr=None, hist_cols=None):
        """(values, columns) [U,k] of ``full_sort_predict`` after recbole's evaluation mask, without the [U, N] matrix."""
        user_e = F_.gather_rows(self.user_embedding.weight, interaction[self.TARGET_USER_ID])
        return F_.fullsort_topk(user_e, self.item_embedding.weight[:self.target_num_items], None, k=k,
                                hist_indptr=hist_indptr, hist_cols=hist_cols, exclude_first_col=True)
