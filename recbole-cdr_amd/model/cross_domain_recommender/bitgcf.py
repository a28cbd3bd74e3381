"""BiTGCF on libcdrhip -- same class contract as recbole_cdr/model/cross_domain_recommender/bitgcf.py:25-282.

The reference propagates over the FULL graph inside every calculate_loss (bitgcf.py:209): 2 domains x n_layers x
[torch.sparse.mm + 3 elementwise ops + transfer (a dozen slices/cats) + normalize].  Here that is one autograd node
(functional.BiTGCFPropagate) of CSR SpMM kernels with the layer math in the epilogue, one transfer kernel per row block
and a normalise-into-the-stack kernel; the per-batch part is the same fused gather-dot-BCE kernel CMF uses, with the
EmbLoss taken on the ego rows.  Adjacency values are formed exactly as the reference does (float64 D^-1/2 A D^-1/2 with
degree + 1e-7, rounded to fp32: bitgcf.py:92-116) -- golden-pinned bit for bit in tests/test_oracle_golden.py.
Dropout (bitgcf.py:66,134): identity in eval; in training a native counter-based mask (cdr_dropout_dev) whose seed sits in device
memory: drawn from torch's CPU generator per step when run eagerly (``torch.manual_seed`` makes runs repeatable), advanced by a
captured kernel when the step is replayed as a hipGraph (a fresh mask per replay).  Masks are not bit-comparable with
the reference's Philox stream: golden parity is pinned at drop_rate = 0, the mask by its statistics.
"""
import numpy as np
import torch
import torch.nn as nn

from ... import binding as B_
from ... import functional as F_
from ...utils import InputType
from ..crossdomain_recommender import CrossDomainRecommender, xavier_normal_initialization


def _csr_norm_adj(pairs, n_users, n_items, device):
    """D^-1/2 A D^-1/2 of the bipartite graph as a device CSR, BUILT ON THE DEVICE (bitgcf.py:92-116 goes through a scipy DOK
    matrix on the host): de-duplication and the (row, column) ordering are device sorts of integer keys, degrees a
    device bincount.  The values stay bit-identical to the reference's: the n degree^-1/2 factors are formed in float64 with the
    host's libm (np.power, as the reference does; n numbers), the per-edge product ((d_i^-1/2 * 1.0) * d_j^-1/2) is an IEEE
    float64 multiply on the device, rounded to fp32 once -- golden-pinned in tests/test_oracle_golden.py / test_gpu_parity.py."""
    p = torch.as_tensor(np.asarray(pairs, dtype=np.int64) if not torch.is_tensor(pairs) else pairs).to(device)
    n = n_users + n_items
    key = torch.unique(p[:, 0] * n_items + p[:, 1])                       # distinct (user, item) edges, sorted
    u, i = key // n_items, key % n_items + n_users
    row, col = torch.cat([u, i]), torch.cat([i, u])
    deg = torch.bincount(row, minlength=n)
    dinv = torch.from_numpy(np.power(deg.cpu().numpy().astype(np.float64) + 1e-7, -0.5)).to(device)
    val = ((dinv[row] * 1.0) * dinv[col]).to(torch.float32)
    order = torch.argsort(row * n + col)                                  # CSR order: by row, columns ascending
    indptr = torch.zeros(n + 1, device=device, dtype=torch.int64)
    indptr[1:] = torch.cumsum(deg, 0)
    return F_.CSRGraph(indptr, col[order].contiguous(), val[order].contiguous(), n)


class BiTGCF(CrossDomainRecommender):
    input_type = InputType.POINTWISE

    def __init__(self, config, dataset):
        super().__init__(config, dataset)
        self.SOURCE_LABEL = dataset.source_domain_dataset.label_field
        self.TARGET_LABEL = dataset.target_domain_dataset.label_field
        self.latent_dim = config['embedding_size']
        self.n_layers = config['n_layers']
        self.reg_weight = config['reg_weight']
        self.domain_lambda_source = config['lambda_source']
        self.domain_lambda_target = config['lambda_target']
        self.drop_rate = config['drop_rate']
        self.connect_way = config['connect_way']
        # config['bitgcf_sparse_last_layer'] = False: every row of the last layer is computed in calculate_loss, as the reference does
        self.sparse_last_layer = bool(config['bitgcf_sparse_last_layer']) if 'bitgcf_sparse_last_layer' in config else True
        # config['bitgcf_fused_loss'] = False: propagation, point loss and EmbLoss as separate autograd nodes (the round-4 form)
        self.fused_loss = bool(config['bitgcf_fused_loss']) if 'bitgcf_fused_loss' in config else True

        self.source_user_embedding = nn.Embedding(self.total_num_users, self.latent_dim)
        self.target_user_embedding = nn.Embedding(self.total_num_users, self.latent_dim)
        self.source_item_embedding = nn.Embedding(self.total_num_items, self.latent_dim)
        self.target_item_embedding = nn.Embedding(self.total_num_items, self.latent_dim)

        dev = self.device
        s_m = dataset.inter_matrix(form='coo', value_field=None, domain='source').astype(np.float32)
        t_m = dataset.inter_matrix(form='coo', value_field=None, domain='target').astype(np.float32)
        self.source_graph = _csr_norm_adj(np.stack([s_m.row, s_m.col], 1), self.total_num_users, self.total_num_items, dev)
        self.target_graph = _csr_norm_adj(np.stack([t_m.row, t_m.col], 1), self.total_num_users, self.total_num_items, dev)
        f = lambda a: torch.from_numpy(np.asarray(a, dtype=np.float32).reshape(-1).copy()).to(dev)
        self.degrees = {'su': f(s_m.sum(axis=1)), 'tu': f(t_m.sum(axis=1)), 'si': f(s_m.sum(axis=0)), 'ti': f(t_m.sum(axis=0))}

        self.target_restore_user_e = None
        self.target_restore_item_e = None
        self.apply(xavier_normal_initialization)
        self.other_parameter_name = ['target_restore_user_e', 'target_restore_item_e']

    def _propagate(self, rows_hint=None, emb_loss=False):
        """(S, T, reg_s, reg_t): the propagated [users ; items] stacks of the two domains, one tensor each, and (``emb_loss``) the
        EmbLoss of the two batches' ego rows.  ``rows_hint`` = (user ids, item ids, ...): the only rows the caller will read --
        the last layer is then evaluated on those rows alone (see BiTGCFPropagate)."""
        return F_.BiTGCFPropagate.apply(self.source_user_embedding.weight, self.source_item_embedding.weight,
                                        self.target_user_embedding.weight, self.target_item_embedding.weight,
                                        self.source_graph, self.target_graph, self.degrees, int(self.n_layers),
                                        float(self.domain_lambda_source), float(self.domain_lambda_target),
                                        self.connect_way, int(self.overlapped_num_users), int(self.overlapped_num_items),
                                        *self._dropout_args(), rows_hint, emb_loss)

    def forward(self):
        S, T, _, _ = self._propagate()
        nu = self.total_num_users
        return S[:nu], S[nu:], T[:nu], T[nu:]

    def _dropout_args(self):
        """(p, seed).  The seed lives in a DEVICE int64: eagerly it is re-drawn from torch's CPU generator on every training forward
        (``torch.manual_seed`` makes a step repeatable); inside a hipGraph capture (graph_step.GraphedTrainStep) a host draw would be
        baked into the graph and every replay would repeat ONE mask, so there the captured step advances the device value itself
        with one tiny kernel and each replay draws a fresh mask, as nn.Dropout does."""
        if not self.training or not self.drop_rate:
            return 0.0, 0
        dev = self.source_user_embedding.weight.device
        st = self.__dict__.get('_drop_state')
        if st is None or st.device != dev:
            st = torch.zeros(1, device=dev, dtype=torch.int64)
            self.__dict__['_drop_state'] = st
        if torch.cuda.is_current_stream_capturing():
            B_.call('cdr_inc_i64', B_.stream(), B_.i64(st))
        else:
            st.fill_(int(torch.empty((), dtype=torch.int64).random_(0, 2 ** 62).item()))
        return float(self.drop_rate), st

    def graph_key(self):
        return ('BiTGCF',)          # (the dropout seed is a device value the captured step advances itself: _dropout_args)

    def calculate_loss(self, interaction):
        self.init_restore_e()
        # the loss reads the batch's user and item rows of the two stacks and nothing else (the transfer couples a row of one domain
        # with the SAME row of the other, so both batches flag both stacks)
        hint = (interaction[self.SOURCE_USER_ID], interaction[self.SOURCE_ITEM_ID], interaction[self.TARGET_USER_ID],
                interaction[self.TARGET_ITEM_ID]) if self.sparse_last_layer else None
        D = self.source_user_embedding.weight.shape[1]
        # (under set_deterministic the one-node form needs its four scatter lists inside the ordered launch's reach)
        det_ok = not F_.deterministic() or F_.ordered_fits((int(self.n_layers) + 1) * D, interaction[self.SOURCE_USER_ID].numel(),
                                                           interaction[self.TARGET_USER_ID].numel())
        if hint is not None and self.fused_loss and D % 4 == 0 and det_ok:
            # propagation + both domains' BCE + reg_weight x EmbLoss as ONE autograd node (functional.BiTGCFLoss): 27 launches per step
            # at BASELINE C4 instead of 37
            return F_.BiTGCFLoss.apply(self.source_user_embedding.weight, self.source_item_embedding.weight,
                                       self.target_user_embedding.weight, self.target_item_embedding.weight,
                                       self.source_graph, self.target_graph, self.degrees, int(self.n_layers),
                                       float(self.domain_lambda_source), float(self.domain_lambda_target), self.connect_way,
                                       int(self.overlapped_num_users), int(self.overlapped_num_items), *self._dropout_args(), float(self.reg_weight),
                                       interaction[self.SOURCE_USER_ID], interaction[self.SOURCE_ITEM_ID], interaction[self.SOURCE_LABEL],
                                       interaction[self.TARGET_USER_ID], interaction[self.TARGET_ITEM_ID], interaction[self.TARGET_LABEL])
        S, T, reg_s, reg_t = self._propagate(hint, emb_loss=hint is not None)
        nu = self.total_num_users
        su, si, sl = interaction[self.SOURCE_USER_ID], interaction[self.SOURCE_ITEM_ID], interaction[self.SOURCE_LABEL]
        tu, ti, tl = interaction[self.TARGET_USER_ID], interaction[self.TARGET_ITEM_ID], interaction[self.TARGET_LABEL]
        # rows of the stacked [users ; items] tables (items at row nu + id): no slices, so each loss's gradient is ONE buffer of the
        # stack's shape handed straight to the propagation's backward; both domains' losses in one launch each way when the width allows
        if S.shape[1] % 4 == 0 and S.is_contiguous() and T.is_contiguous():
            bce_s, bce_t = F_.TwoStackPointLoss.apply(B_.CDR_LOSS_BCE, S, T, nu, su, si, sl, tu, ti, tl)
        else:
            bce_s, _ = F_.PointGatherLoss.apply(B_.CDR_LOSS_BCE, S, S, None, None, su, si + nu, sl, 0.0)
            bce_t, _ = F_.PointGatherLoss.apply(B_.CDR_LOSS_BCE, T, T, None, None, tu, ti + nu, tl, 0.0)
        losses = []
        for bce, uw, iw, user, item, reg in ((bce_s, self.source_user_embedding.weight, self.source_item_embedding.weight, su, si, reg_s),
                                             (bce_t, self.target_user_embedding.weight, self.target_item_embedding.weight, tu, ti, reg_t)):
            if reg is None:
                reg = F_.EmbLossRows.apply(uw, iw, user, item)
            losses.append(torch.add(bce, reg, alpha=self.reg_weight))        # bce + reg_weight * reg in ONE elementwise launch (a 4.7 us launch each at C4)
        return tuple(losses)

    @torch.no_grad()
    def predict(self, interaction):
        _, _, tu_all, ti_all = self.forward()
        user, item = interaction[self.TARGET_USER_ID], interaction[self.TARGET_ITEM_ID]
        zeros = torch.zeros(user.numel(), device=tu_all.device, dtype=torch.float32)
        _, scores = F_.PointGatherLoss.apply(B_.CDR_LOSS_MSE, tu_all.contiguous(), ti_all.contiguous(), None, None, user, item,
                                             zeros, 0.0)
        return scores

    @torch.no_grad()
    def full_sort_predict(self, interaction):
        restore_user_e, restore_item_e = self.get_restore_e()
        u = F_.gather_rows(restore_user_e, interaction[self.TARGET_USER_ID])
        return F_.fullsort_scores(u, restore_item_e[:self.target_num_items]).view(-1)

    @torch.no_grad()
    def full_sort_topk(self, interaction, k, hist_indptr=None, hist_cols=None):
        """(values, columns) [U,k] of ``full_sort_predict`` after recbole's evaluation mask, without the [U, N] matrix."""
        restore_user_e, restore_item_e = self.get_restore_e()
        u = F_.gather_rows(restore_user_e, interaction[self.TARGET_USER_ID])
        return F_.fullsort_topk(u, restore_item_e[:self.target_num_items], None, k=k, hist_indptr=hist_indptr,
                                hist_cols=hist_cols, exclude_first_col=True)

    def on_train_steps(self):
        # a replayed step runs no Python: the trainer calls this after replays so that the next evaluation propagates the TRAINED tables
        self.init_restore_e()

    def train(self, mode=True):
        if mode:
            self.init_restore_e()                 # (whoever trains without this package's Trainer still gets a fresh cache per training visit)
        return super().train(mode)

    def init_restore_e(self):
        if self.target_restore_user_e is not None or self.target_restore_item_e is not None:
            self.target_restore_user_e, self.target_restore_item_e = None, None

    @torch.no_grad()
    def get_restore_e(self):
        if self.target_restore_user_e is None or self.target_restore_item_e is None:
            _, _, tu, ti = self.forward()
            self.target_restore_user_e, self.target_restore_item_e = tu.contiguous(), ti.contiguous()
        return self.target_restore_user_e, self.target_restore_item_e
