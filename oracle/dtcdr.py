"""DTCDR (base_model = NeuMF) restated (oracle; test infrastructure only).  /root/reference
recbole_cdr/model/cross_domain_recommender/dtcdr.py: neumf_forward :112-126, calculate_loss :182-199, predict :201-207.
recbole MLPLayers (un-vendored): Dropout -> Linear -> ReLU for every consecutive pair, parameters named
``mlp_layers.{1,4,7,...}``.  Dropout is the identity here (p = 0 / eval) unless explicit masks are handed in (``masks[(domain,
layer index)]``, already scaled by 1 / (1 - p): the mask itself is an RNG stream; with the product's mask both sides compute the
same function).
The -inf fill of dtcdr.py:54-59 is overwritten by ``self.apply(xavier_normal_initialization)`` (:107) and has no effect."""
import torch

from .losses import bce_loss


def _mlp(params, prefix, x, masks=None, domain=None):
    n = 1
    while f'{prefix}.mlp_layers.{n}.weight' in params:
        if masks is not None:
            x = x * masks[(domain, (n - 1) // 3)]                 # recbole MLPLayers: Dropout in front of every Linear
        x = torch.relu(x @ params[f'{prefix}.mlp_layers.{n}.weight'].t() + params[f'{prefix}.mlp_layers.{n}.bias'])
        n += 3
    return x


def neumf_forward(params, user, item, domain, masks=None):
    user_e = torch.maximum(params['source_user_embedding.weight'][user], params['target_user_embedding.weight'][user])
    item_e = torch.maximum(params['source_item_embedding.weight'][item], params['target_item_embedding.weight'][item])
    h = _mlp(params, f'{domain}_mlp_layers', torch.cat((user_e, item_e), -1), masks, domain)
    out = torch.sigmoid(h @ params[f'{domain}_predict_layer.weight'].t() + params[f'{domain}_predict_layer.bias'])
    return out.squeeze(-1)


def calculate_loss(params, ids, inter, alpha, masks=None):
    ls = bce_loss(neumf_forward(params, inter['source_user_id'], inter['source_item_id'], 'source', masks), inter['source_label'])
    lt = bce_loss(neumf_forward(params, inter['target_user_id'], inter['target_item_id'], 'target', masks), inter['target_label'])
    return ls * alpha + lt * (1 - alpha)


def predict(params, ids, inter):
    return neumf_forward(params, inter['target_user_id'], inter['target_item_id'], 'target')
