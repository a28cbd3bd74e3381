"""DCDCSR on libcdrhip -- same class contract as recbole_cdr/model/cross_domain_recommender/dcdcsr.py:25-280 (trained by
DCDCSRTrainer: SOURCE -> TARGET -> BOTH -> TARGET).

* SOURCE / first TARGET (dcdcsr.py:112-127): BPR without a regulariser -- the fused gather-dot-BPR kernel with reg_weight 0.
* set_phase('BOTH') builds the benchmark embedding (dcdcsr.py:129-165).  The reference loops over every unit in Python with a
  [n_overlap] matmul + topk per unit; here the similarity top-k of ALL non-overlapped units is one call of the fused MFMA
  scoring + top-k kernel (cdr_fullsort_topk_f32: the [units, n_overlap] similarity matrix is never written) followed by a few
  [units, k] elementwise ops -- once per phase switch, not per step.
* BOTH (dcdcsr.py:174-182): numpy-sampled unit ids -> gather -> max-min row normalisation (cdr_maxmin_norm, native backward) ->
  tanh MLP on the MFMA contraction -> MSE against the normalised benchmark rows.
* second TARGET: BPR with the detached affine table standing in for the overlapped side's table (dcdcsr.py:98-110, 204-213).
``predict`` / ``full_sort_predict`` follow the reference's phase table, including full_sort_predict returning [U, N]."""
import numpy as np
import torch
import torch.nn as nn

from ... import binding as B_
from ... import functional as F_
from ...utils import InputType
from ..crossdomain_recommender import CrossDomainRecommender, xavier_normal_initialization
from .sscdr import MLPLayers


class DCDCSR(CrossDomainRecommender):
    input_type = InputType.PAIRWISE

    def __init__(self, config, dataset):
        super().__init__(config, dataset)
        self.mode = self.one_sided_overlap_mode()
        self.phase = None
        self.phase2count = {'SOURCE': 0, 'TARGET': 0, 'BOTH': 0, 'OVERLAP': 0}
        self.latent_factor_model = config['latent_factor_model']
        assert self.latent_factor_model in ['BPR'], f'DCDCSR trains its latent factors with BPR only (latent_factor_model={self.latent_factor_model!r})'
        self.embedding_size = config['embedding_size']
        self.mlp_hidden_size = list(config['mlp_hidden_size'])
        self.k = config['k']
        self.map_batch_size = config['map_batch_size']
        self.SOURCE_LABEL = dataset.source_domain_dataset.label_field
        self.TARGET_LABEL = dataset.target_domain_dataset.label_field
        if self.mode == 'overlap_items':
            self.source_item2pop = self.build_unit2pop(dataset, unit='item', domain='source').to(self.device)
            self.target_item2pop = self.build_unit2pop(dataset, unit='item', domain='target').to(self.device)
        elif self.mode == 'overlap_users':
            self.source_user2pop = self.build_unit2pop(dataset, unit='user', domain='source').to(self.device)
            self.target_user2pop = self.build_unit2pop(dataset, unit='user', domain='target').to(self.device)

        self.source_user_embedding = nn.Embedding(self.total_num_users, self.embedding_size)
        self.source_item_embedding = nn.Embedding(self.total_num_items, self.embedding_size)
        self.target_user_embedding = nn.Embedding(self.total_num_users, self.embedding_size)
        self.target_item_embedding = nn.Embedding(self.total_num_items, self.embedding_size)
        self.benchmark_embedding = None
        self.affine_embedding = None
        # (zero fills of dcdcsr.py:72-76 are overwritten by the initialisation below, :86)
        self.mapping_mlp_layers = MLPLayers([self.embedding_size] + self.mlp_hidden_size + [self.embedding_size])
        self.apply(xavier_normal_initialization)

    @staticmethod
    def build_unit2pop(dataset, unit='user', domain='source'):
        if unit == 'user':
            _, _, history_lens = dataset.history_item_matrix(domain=domain)
        else:
            _, _, history_lens = dataset.history_user_matrix(domain=domain)
        return history_lens.float()

    def _unit(self):
        return 'user' if self.mode == 'overlap_users' else 'item'

    def set_phase(self, phase):
        self.phase = phase
        self.phase2count[phase] += 1
        if self.phase == 'BOTH':
            self.build_benchmark_embedding()
        if self.phase == 'TARGET' and self.phase2count[self.phase] == 2:
            unit = self._unit()
            n_tgt = getattr(self, f'target_num_{unit}s')
            with torch.no_grad():
                e, stats = F_.MaxMinNormalize.apply(getattr(self, f'target_{unit}_embedding').weight[:n_tgt].detach())
                mean_, max_ = stats[:, :1], stats[:, 1:]
                self.affine_embedding = (self.mapping_mlp_layers(e) * (max_ - mean_) + mean_).detach()

    @torch.no_grad()
    def build_unit_benchmark_embedding(self, total_num_units, overlapped_num_units, source_unit2pop, target_unit2pop,
                                       source_unit_embeddings, target_unit_embeddings):
        src = source_unit_embeddings.detach().contiguous()
        tgt = target_unit_embeddings.detach()
        no = overlapped_num_units
        bench = torch.empty(total_num_units, self.embedding_size, device=tgt.device, dtype=torch.float32)
        den = source_unit2pop[:no] + target_unit2pop[:no]
        den = torch.where(den == 0, torch.ones_like(den), den)
        a_s = (source_unit2pop[:no] / den).unsqueeze(1)
        bench[:no] = a_s * tgt[:no] + (1 - a_s) * src
        if total_num_units > no:
            rest = tgt[no:].contiguous()
            sim, index = F_.fullsort_topk(rest, src, None, k=self.k, exclude_first_col=False)        # [units, k] each
            sn = source_unit2pop[index].mean(dim=1)
            beta = (sn / (sn + target_unit2pop[no:])).unsqueeze(1)
            sim_e = torch.bmm(sim.unsqueeze(1), src[index]).squeeze(1)                                 # [units, D]
            sum_sim = sim.sum(dim=1, keepdim=True)
            sum_sim = torch.where(sum_sim > 0, sum_sim, torch.ones_like(sum_sim))
            bench[no:] = (1 - beta) * rest + beta * (sim_e / sum_sim)
        self.benchmark_embedding = bench

    def build_benchmark_embedding(self):
        if self.mode == 'overlap_users':
            self.build_unit_benchmark_embedding(self.total_num_users, self.overlapped_num_users, self.source_user2pop,
                                                self.target_user2pop,
                                                self.source_user_embedding.weight[:self.overlapped_num_users],
                                                self.target_user_embedding.weight)
        elif self.mode == 'overlap_items':
            self.build_unit_benchmark_embedding(self.total_num_items, self.overlapped_num_items, self.source_item2pop,
                                                self.target_item2pop,
                                                self.source_item_embedding.weight[:self.overlapped_num_items],
                                                self.target_item_embedding.weight)

    def maxmin_normalize(self, embed_weight):
        y, stats = F_.MaxMinNormalize.apply(embed_weight)
        return y, stats[:, :1], stats[:, 1:]

    def calculate_rec_loss(self, interaction, user_embeds, item_embeds, user_field, item_field, neg_item_field, label_field):
        return F_.BPRGatherLoss.apply(user_embeds, item_embeds, interaction[user_field], interaction[item_field],
                                      interaction[neg_item_field], 1e-10, 0.0).reshape(())

    def calculate_unit_map_loss(self, target_num_units, target_unit_embeddings):
        sampled_index = np.random.randint(0, target_num_units, self.map_batch_size)        # host draw, as the reference (:175)
        idx = torch.from_numpy(sampled_index).to(target_unit_embeddings.device)
        e, _, _ = self.maxmin_normalize(F_.gather_rows(target_unit_embeddings, idx))
        mapped = self.mapping_mlp_layers(e)
        with torch.no_grad():
            b, _, _ = self.maxmin_normalize(F_.gather_rows(self.benchmark_embedding, idx))
        return F_.mse_loss(mapped, b)

    def calculate_map_loss(self):
        if self.mode == 'overlap_users':
            return self.calculate_unit_map_loss(self.target_num_users, self.target_user_embedding.weight)
        elif self.mode == 'overlap_items':
            return self.calculate_unit_map_loss(self.target_num_items, self.target_item_embedding.weight)
        return None

    def _tables(self):
        """(user table, item table, domain tag) of the current phase -- the branch table shared by calculate_loss :192-213,
        full_sort_predict :215-245 and predict :247-280."""
        first = self.phase2count.get(self.phase, 0) == 1
        if self.phase == 'SOURCE' and first:
            return self.source_user_embedding.weight, self.source_item_embedding.weight, 'SOURCE'
        if self.phase == 'TARGET' and first:
            return self.target_user_embedding.weight, self.target_item_embedding.weight, 'TARGET'
        if self.mode == 'overlap_users':
            return self.affine_embedding, self.target_item_embedding.weight, 'TARGET'
        return self.target_user_embedding.weight, self.affine_embedding, 'TARGET'

    def graph_key(self):
        # BOTH: the map loss draws its rows with numpy on the host (:175 of the reference) -- not capturable.  The other phases read
        # tables that depend on how often the phase has been visited (affine_embedding is rebuilt on the second TARGET visit).
        return None if self.phase == 'BOTH' else ('DCDCSR', self.phase, self.phase2count.get(self.phase, 0))

    def calculate_loss(self, interaction):
        count = self.phase2count.get(self.phase, 0)
        if self.phase == 'BOTH':
            return self.calculate_map_loss()
        if (self.phase == 'SOURCE' and count == 1) or (self.phase == 'TARGET' and count in (1, 2)):
            U, I, tag = self._tables()
            return self.calculate_rec_loss(interaction, U, I, getattr(self, f'{tag}_USER_ID'), getattr(self, f'{tag}_ITEM_ID'),
                                           getattr(self, f'{tag}_NEG_ITEM_ID'), getattr(self, f'{tag}_LABEL'))
        return None

    @torch.no_grad()
    def full_sort_predict(self, interaction):
        U, I, tag = self._tables()
        user_e = F_.gather_rows(U, interaction[getattr(self, f'{tag}_USER_ID')])
        if tag == 'SOURCE':
            return F_.fullsort_scores(user_e, I[:self.overlapped_num_items], I[self.target_num_items:])
        if I is self.affine_embedding:
            return F_.fullsort_scores(user_e, I)
        return F_.fullsort_scores(user_e, I[:self.target_num_items])

    @torch.no_grad()
    def predict(self, interaction):
        U, I, tag = self._tables()
        user, item = interaction[getattr(self, f'{tag}_USER_ID')], interaction[getattr(self, f'{tag}_ITEM_ID')]
        zeros = torch.zeros(user.numel(), device=U.device, dtype=torch.float32)
        return F_.PointGatherLoss.apply(B_.CDR_LOSS_MSE, U.contiguous(), I.contiguous(), None, None, user, item, zeros, 0.0)[1]
