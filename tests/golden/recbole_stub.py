"""Stand-in for the un-vendored third-party dependency ``recbole==1.0.1``.

GENERATOR-SIDE ONLY.  This file is used by ``tests/golden/make_golden.py`` in
the build container to import the reference's model classes from
``/root/reference`` (read-only) and dump numeric golden vectors.  It is never
imported by the product, by ``tests/test_*.py``, by ``bench.py`` or by
``__graft_entry__``.

The reference pins ``recbole==1.0.1`` (requirements.txt:1) which is neither
vendored under /root/reference nor installed in this image.  Seven symbols
carry arithmetic on the hot path; they are restated here from the published
recbole 1.0.1 semantics.  Everything else is an inert
placeholder that only has to exist at import time.

PARITY NOTE: no test of the reference pins these seven symbols, so fixtures
that flow through them (BPRLoss, EmbLoss, MLPLayers, xavier init) are
"parity unpinned" at the recbole boundary; all stock-torch arithmetic
(MSELoss, BCELoss, TripletMarginLoss, Embedding, Linear, matmul, sparse.mm)
is exercised for real because torch is present.
"""
import sys
import types
from enum import Enum

import torch
import torch.nn as nn
from torch.nn.init import xavier_normal_, constant_


# --------------------------------------------------------------------------- arithmetic symbols
class InputType(Enum):
    POINTWISE = 1
    PAIRWISE = 2
    LISTWISE = 3


class AbstractRecommender(nn.Module):
    def __init__(self):
        super().__init__()

    def calculate_loss(self, interaction):
        raise NotImplementedError

    def predict(self, interaction):
        raise NotImplementedError

    def full_sort_predict(self, interaction):
        raise NotImplementedError

    def other_parameter(self):
        if hasattr(self, 'other_parameter_name'):
            return {key: getattr(self, key) for key in self.other_parameter_name}
        return dict()

    def load_other_parameter(self, para):
        if para is None:
            return
        for key, value in para.items():
            setattr(self, key, value)


def xavier_normal_initialization(module):
    if isinstance(module, nn.Embedding):
        xavier_normal_(module.weight.data)
    elif isinstance(module, nn.Linear):
        xavier_normal_(module.weight.data)
        if module.bias is not None:
            constant_(module.bias.data, 0)


class BPRLoss(nn.Module):
    def __init__(self, gamma=1e-10):
        super().__init__()
        self.gamma = gamma

    def forward(self, pos_score, neg_score):
        return -torch.log(self.gamma + torch.sigmoid(pos_score - neg_score)).mean()


class EmbLoss(nn.Module):
    def __init__(self, norm=2):
        super().__init__()
        self.norm = norm

    def forward(self, *embeddings, require_pow=False):
        if require_pow:
            emb_loss = torch.zeros(1).to(embeddings[-1].device)
            for embedding in embeddings:
                emb_loss += torch.pow(input=torch.norm(embedding, p=self.norm), exponent=self.norm)
            emb_loss /= embeddings[-1].shape[0]
            emb_loss /= self.norm
            return emb_loss
        emb_loss = torch.zeros(1).to(embeddings[-1].device)
        for embedding in embeddings:
            emb_loss += torch.norm(embedding, p=self.norm)
        emb_loss /= embeddings[-1].shape[0]
        return emb_loss


class RegLoss(nn.Module):
    def forward(self, parameters):
        reg_loss = None
        for W in parameters:
            reg_loss = W.norm(2) if reg_loss is None else reg_loss + W.norm(2)
        return reg_loss


class MLPLayers(nn.Module):
    """Dropout -> Linear -> [BN] -> activation for EVERY consecutive pair (also the last)."""

    def __init__(self, layers, dropout=0., activation='relu', bn=False, init_method=None):
        super().__init__()
        mods = []
        for d_in, d_out in zip(layers[:-1], layers[1:]):
            mods.append(nn.Dropout(p=dropout))
            mods.append(nn.Linear(d_in, d_out))
            if bn:
                mods.append(nn.BatchNorm1d(num_features=d_out))
            act = {'relu': nn.ReLU, 'tanh': nn.Tanh, 'sigmoid': nn.Sigmoid,
                   'leakyrelu': nn.LeakyReLU, 'none': None}[activation.lower()]
            if act is not None:
                mods.append(act())
        self.mlp_layers = nn.Sequential(*mods)

    def forward(self, x):
        return self.mlp_layers(x)


# --------------------------------------------------------------------------- inert placeholders
class _Inert:
    def __init__(self, *a, **k):
        pass


class ModelTypeStub(Enum):
    GENERAL = 1


class _EnumLike(Enum):
    TOKEN = 'token'
    FLOAT = 'float'
    TOKEN_SEQ = 'token_seq'
    FLOAT_SEQ = 'float_seq'


class FeatureSource(Enum):
    INTERACTION = 'inter'
    USER = 'user'
    ITEM = 'item'
    USER_ID = 'user_id'
    ITEM_ID = 'item_id'
    KG = 'kg'
    NET = 'net'


class EvaluatorType(Enum):
    RANKING = 1
    VALUE = 2


def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def install():
    """Register the stub package tree in ``sys.modules`` (idempotent)."""
    if 'recbole' in sys.modules and getattr(sys.modules['recbole'], '_IS_STUB', False):
        return
    noop = lambda *a, **k: None
    Interaction = type('Interaction', (dict,), {})
    _mod('recbole', _IS_STUB=True)
    _mod('recbole.utils', InputType=InputType, ModelType=ModelTypeStub, EvaluatorType=EvaluatorType,
         FeatureSource=FeatureSource, FeatureType=_EnumLike, init_logger=noop, init_seed=noop,
         set_color=lambda s, *a, **k: s, get_model=noop, get_trainer=noop)
    _mod('recbole.utils.argument_list', dataset_arguments=[])
    _mod('recbole.config')
    _mod('recbole.config.configurator', Config=_Inert)
    _mod('recbole.evaluator', metric_types={}, smaller_metrics=[])
    _mod('recbole.data')
    _mod('recbole.data.dataset', Dataset=_Inert)
    _mod('recbole.data.interaction', Interaction=Interaction)
    _mod('recbole.data.utils', load_split_dataloaders=noop, save_split_dataloaders=noop, create_samplers=noop)
    _mod('recbole.data.dataloader', NegSampleEvalDataLoader=_Inert)
    _mod('recbole.data.dataloader.abstract_dataloader', AbstractDataLoader=_Inert)
    _mod('recbole.data.dataloader.general_dataloader', TrainDataLoader=_Inert, FullSortEvalDataLoader=_Inert)
    _mod('recbole.trainer', Trainer=_Inert, HyperTuning=_Inert)
    _mod('recbole.model')
    _mod('recbole.model.abstract_recommender', AbstractRecommender=AbstractRecommender)
    _mod('recbole.model.init', xavier_normal_initialization=xavier_normal_initialization)
    _mod('recbole.model.loss', BPRLoss=BPRLoss, EmbLoss=EmbLoss, RegLoss=RegLoss)
    _mod('recbole.model.layers', MLPLayers=MLPLayers)
