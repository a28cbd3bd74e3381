"""EMCDR restated (oracle; test infrastructure only).  Follows /root/reference
recbole_cdr/model/cross_domain_recommender/emcdr.py line by line:
  source/target_forward :98-108, calculate_source_loss :110-131, calculate_target_loss :133-154,
  calculate_map_loss :156-168, calculate_loss :170-176, predict :178-206, full_sort_predict :208-233,
  mapping = Linear(bias=False) :59-60 or Linear->Tanh->...->Linear (no final act) :86-93.

``params`` is a dict keyed like the reference's ``named_parameters()``:
  source_user_embedding.weight, source_item_embedding.weight, target_user_embedding.weight,
  target_item_embedding.weight, mapping.weight (linear) | mapping.{0,2,..}.{weight,bias} (non_linear).
"""
import torch
import torch.nn.functional as F

from .losses import bpr_loss, emb_loss, mse_loss


def mapping(params, x):
    if 'mapping.weight' in params:
        return F.linear(x, params['mapping.weight'])
    idx = sorted({int(k.split('.')[1]) for k in params if k.startswith('mapping.')})
    for n, i in enumerate(idx):
        x = F.linear(x, params[f'mapping.{i}.weight'], params[f'mapping.{i}.bias'])
        if n != len(idx) - 1:
            x = torch.tanh(x)
    return x


def _dot(ue, ie):
    return torch.mul(ue, ie).sum(dim=1)


def domain_loss(params, inter, domain, latent_factor_model, reg_weight):
    """calculate_source_loss / calculate_target_loss (emcdr.py:110-154)."""
    U = params[f'{domain}_user_embedding.weight']
    I = params[f'{domain}_item_embedding.weight']
    user = inter[f'{domain}_user_id']
    item = inter[f'{domain}_item_id']
    if latent_factor_model == 'MF':
        label = inter[f'{domain}_label']
        p = _dot(U[user], I[item])
        return mse_loss(p, label) + reg_weight * emb_loss(U[user], I[item])
    neg = inter[f'neg_{domain}_item_id']
    pos_score = _dot(U[user], I[item])
    neg_score = _dot(U[user], I[neg])
    return bpr_loss(pos_score, neg_score) + reg_weight * emb_loss(U[user], I[item])


def map_loss(params, ids, inter):
    idx = inter['overlap']                     # [OB,1]
    kind = 'user' if ids.mode == 'overlap_users' else 'item'
    src = params[f'source_{kind}_embedding.weight'][idx]
    tgt = params[f'target_{kind}_embedding.weight'][idx]
    return mse_loss(mapping(params, src), tgt)


def calculate_loss(params, ids, inter, phase, latent_factor_model='MF', reg_weight=0.01):
    if phase == 'SOURCE':
        return domain_loss(params, inter, 'source', latent_factor_model, reg_weight)
    if phase == 'OVERLAP':
        return map_loss(params, ids, inter)
    return domain_loss(params, inter, 'target', latent_factor_model, reg_weight)


def _mapped_users(params, ids, user):
    D = params['source_user_embedding.weight'].shape[1]
    rep = user.repeat(D, 1).transpose(0, 1)
    return torch.where(rep < ids.OU, mapping(params, params['source_user_embedding.weight'][user]),
                       params['target_user_embedding.weight'][user])


def predict(params, ids, inter, phase):
    if phase == 'SOURCE':
        return _dot(params['source_user_embedding.weight'][inter['source_user_id']],
                    params['source_item_embedding.weight'][inter['source_item_id']])
    user, item = inter['target_user_id'], inter['target_item_id']
    if phase == 'TARGET':
        return _dot(params['target_user_embedding.weight'][user], params['target_item_embedding.weight'][item])
    if ids.mode == 'overlap_users':
        ue = _mapped_users(params, ids, user)
        ie = params['target_item_embedding.weight'][item]
    else:
        ue = params['target_user_embedding.weight'][user]
        D = params['source_item_embedding.weight'].shape[1]
        rep = item.repeat(D, 1).transpose(0, 1)
        ie = torch.where(rep < ids.OI, mapping(params, params['source_item_embedding.weight'][item]),
                         params['target_item_embedding.weight'][item])
    return _dot(ue, ie)


def full_sort_predict(params, ids, inter, phase):
    TI = ids.target_num_items
    if phase == 'SOURCE':
        ue = params['source_user_embedding.weight'][inter['source_user_id']]
        W = params['source_item_embedding.weight']
        all_item = torch.cat([W[:ids.OI], W[TI:]], dim=0)
    elif phase == 'TARGET':
        ue = params['target_user_embedding.weight'][inter['target_user_id']]
        all_item = params['target_item_embedding.weight'][:TI]
    else:
        user = inter['target_user_id']
        if ids.mode == 'overlap_users':
            ue = _mapped_users(params, ids, user)
            all_item = params['target_item_embedding.weight'][:TI]
        else:
            ue = params['target_user_embedding.weight'][user]
            ov = mapping(params, params['source_item_embedding.weight'][:ids.OI])
            all_item = torch.cat([ov, params['target_item_embedding.weight'][ids.OI:TI]], dim=0)
    return torch.matmul(ue, all_item.transpose(0, 1)).view(-1)
