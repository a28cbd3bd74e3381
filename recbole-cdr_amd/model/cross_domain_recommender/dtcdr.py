"""DTCDR on libcdrhip -- same class contract as recbole_cdr/model/cross_domain_recommender/dtcdr.py:23-211, base_model = NeuMF
(the configured default, properties/model/DTCDR.yaml).

NeuMF tower (dtcdr.py:112-126): maximum of the two domains' user rows and of their item rows, written side by side into one
[B, 2D] operand by one gather kernel each (cdr_gather_max2; torch.maximum's tie-splitting backward), recbole ``MLPLayers``
(Dropout -> Linear -> ReLU per layer) on the fp32-MFMA contraction with bias + ReLU in the epilogue, predict layer + sigmoid in
the same kernel, BCE natively.  Dropout: identity in eval; in training the native counter-based mask (functional._dropout) --
not bit-comparable with torch's Philox stream, golden parity is pinned at dropout_prob = 0.
``base_model = 'DMF'`` (dtcdr.py:128-180) is NOT provided: the reference's DMF path indexes the source history with the wrong
batch (``source_history_user_value[user]`` for an item matrix, :156-160), shifts source-only ids with ``>`` instead of ``>=``
(:131,151: id == target_num_items lands outside the [B, source_num_items] matrix whenever target-only >= source-only ids) and
scores the target batch with the source tower (:194); it cannot run on an id space with target-only ids and has no oracle."""
import torch
import torch.nn as nn

from ... import binding as B_
from ... import functional as F_
from ...utils import InputType
from ..crossdomain_recommender import CrossDomainRecommender, xavier_normal_initialization


class MLPLayers(nn.Module):
    """recbole.model.layers.MLPLayers(layers, dropout) with the default ReLU: parameter names ``mlp_layers.<3i+1>.*``."""

    def __init__(self, layers, dropout=0.0):
        super().__init__()
        self.dropout = float(dropout)
        mods = []
        for d_in, d_out in zip(layers[:-1], layers[1:]):
            mods += [nn.Dropout(p=dropout), nn.Linear(d_in, d_out), nn.ReLU()]
        self.mlp_layers = nn.Sequential(*mods)
        self.logger = None

    def forward(self, x, seed=None, salt=0):
        """``seed``: device int64 [1] counter of the step's dropout masks (None: no dropout); layer n uses stream ``salt + n``."""
        n = 0
        for m in self.mlp_layers:
            if isinstance(m, nn.Linear):
                if seed is not None and self.training and self.dropout > 0:
                    x = _Dropout.apply(x, self.dropout, seed, salt + n)
                x = F_.linear(x, m.weight, m.bias, B_.ACT_RELU)
                n += 1
        return x


class _Dropout(torch.autograd.Function):
    """x * mask / (1 - p) with the native counter-based mask (cdr_dropout_dev: the seed is read from device memory, so a captured
    step draws a fresh mask on every replay); the backward re-applies the same mask."""

    @staticmethod
    def forward(ctx, x, p, seed, salt):
        y = x.contiguous().clone()
        F_._dropout(y, y.numel(), p, seed, salt)
        ctx.p, ctx.seed, ctx.salt = p, seed, salt
        return y

    @staticmethod
    def backward(ctx, g):
        g = g.contiguous().clone()
        F_._dropout(g, g.numel(), ctx.p, ctx.seed, ctx.salt)
        return g, None, None, None


class DTCDR(CrossDomainRecommender):
    input_type = InputType.POINTWISE

    def __init__(self, config, dataset):
        super().__init__(config, dataset)
        self.SOURCE_LABEL = dataset.source_domain_dataset.label_field
        self.TARGET_LABEL = dataset.target_domain_dataset.label_field
        self.embedding_size = config['embedding_size']
        self.mlp_hidden_size = list(config['mlp_hidden_size'])
        self.dropout_prob = config['dropout_prob']
        self.base_model = config['base_model']
        self.alpha = config['alpha']
        assert self.base_model in ['NeuMF', 'DMF'], f'DTCDR base_model must be NeuMF or DMF, got {self.base_model!r}'
        if self.base_model != 'NeuMF':
            raise NotImplementedError("DTCDR base_model 'DMF' is not provided by this build (see the module docstring)")

        self.source_user_embedding = nn.Embedding(self.total_num_users, self.embedding_size)
        self.source_item_embedding = nn.Embedding(self.total_num_items, self.embedding_size)
        self.target_user_embedding = nn.Embedding(self.total_num_users, self.embedding_size)
        self.target_item_embedding = nn.Embedding(self.total_num_items, self.embedding_size)
        # the reference fills the rows a domain never sees with -inf here (dtcdr.py:54-59) and then re-initialises every
        # embedding (:107), so the fill has no effect; it is not repeated
        self.source_mlp_layers = MLPLayers([2 * self.embedding_size] + self.mlp_hidden_size, self.dropout_prob)
        self.source_predict_layer = nn.Linear(self.mlp_hidden_size[-1], 1)
        self.target_mlp_layers = MLPLayers([2 * self.embedding_size] + self.mlp_hidden_size, self.dropout_prob)
        self.target_predict_layer = nn.Linear(self.mlp_hidden_size[-1], 1)
        self.apply(xavier_normal_initialization)

    def _drop_seed(self):
        """Device-resident seed of this step's dropout masks: re-drawn from torch's CPU generator when run eagerly, advanced by a
        captured kernel inside a hipGraph capture (a host draw would be baked into the graph: one mask for every replay)."""
        if not self.training or not self.dropout_prob:
            return None
        dev = self.source_user_embedding.weight.device
        st = self.__dict__.get('_drop_state')
        if st is None or st.device != dev:
            st = torch.zeros(1, device=dev, dtype=torch.int64)
            self.__dict__['_drop_state'] = st
        if torch.cuda.is_current_stream_capturing():
            B_.call('cdr_inc_i64', B_.stream(), B_.i64(st))
        else:
            st.fill_(int(torch.empty((), dtype=torch.int64).random_(0, 2 ** 62).item()))
        return st

    def neumf_forward(self, user, item, domain='source', seed=None):
        x = F_.GatherMaxConcat.apply(self.source_user_embedding.weight, self.target_user_embedding.weight,
                                     self.source_item_embedding.weight, self.target_item_embedding.weight, user, item)
        mlp = self.source_mlp_layers if domain == 'source' else self.target_mlp_layers
        head = self.source_predict_layer if domain == 'source' else self.target_predict_layer
        if seed is None:
            seed = self._drop_seed()
        return F_.linear(mlp(x, seed, 0 if domain == 'source' else 64), head.weight, head.bias, B_.ACT_SIGMOID).squeeze(-1)

    def graph_key(self):
        return ('DTCDR',)           # (dropout: a device-side seed the captured step advances itself, _dropout_args)

    def calculate_loss(self, interaction):
        seed = self._drop_seed()
        ps = self.neumf_forward(interaction[self.SOURCE_USER_ID], interaction[self.SOURCE_ITEM_ID], 'source', seed)
        pt = self.neumf_forward(interaction[self.TARGET_USER_ID], interaction[self.TARGET_ITEM_ID], 'target', seed)
        loss_s = F_.BCEProbLoss.apply(ps, interaction[self.SOURCE_LABEL])
        loss_t = F_.BCEProbLoss.apply(pt, interaction[self.TARGET_LABEL])
        return loss_s * self.alpha + loss_t * (1 - self.alpha)

    @torch.no_grad()
    def predict(self, interaction):
        return self.neumf_forward(interaction[self.TARGET_USER_ID], interaction[self.TARGET_ITEM_ID], 'target')
