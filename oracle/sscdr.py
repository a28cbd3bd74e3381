"""SSCDR restated (oracle; test infrastructure only).  /root/reference recbole_cdr/model/cross_domain_recommender/
sscdr.py: sample :89-118, embedding_normalize :120-124, embedding_distance :126-128, source/target loss :133-159,
map loss :161-187, predict :197-226, full_sort_predict :228-259.  mapping_layer = recbole MLPLayers(tanh): Linear+Tanh
for EVERY layer including the last.

params: {source,target}_{user,item}_embedding.weight, mapping_layer.mlp_layers.<1,4,..>.{weight,bias}.
The reference draws the semi-supervised (interacted, non-interacted) ids from the global numpy RNG inside the loss;
the oracle takes them as explicit inputs (``sampled_pos`` / ``sampled_neg``) and offers ``sample`` separately.
"""
import numpy as np
import torch
import torch.nn.functional as F

from .losses import mse_loss, triplet_margin_loss


def mapping_layer(params, x):
    idx = sorted({int(k.split('.')[2]) for k in params if k.startswith('mapping_layer.mlp_layers.')})
    for i in idx:
        x = torch.tanh(F.linear(x, params[f'mapping_layer.mlp_layers.{i}.weight'], params[f'mapping_layer.mlp_layers.{i}.bias']))
    return x


def embedding_normalize(e):
    """Divides by the SQUARED length when it exceeds 1 (sscdr.py:120-124) -- quirk kept."""
    length = torch.sum(e ** 2, dim=1, keepdim=True)
    norm = torch.where(length > 1, length, torch.ones_like(length))
    return e / norm


def embedding_distance(a, b):
    return torch.sum((a - b) ** 2, dim=1)


def domain_loss(params, inter, domain, margin):
    U = params[f'{domain}_user_embedding.weight']
    I = params[f'{domain}_item_embedding.weight']
    u, p, n = inter[f'{domain}_user_id'], inter[f'{domain}_item_id'], inter[f'neg_{domain}_item_id']
    return triplet_margin_loss(embedding_normalize(U[u]), embedding_normalize(I[p]), embedding_normalize(I[n]), margin)


def sample(ids, idspace, hist_lists, mode, rng=np.random):
    """sscdr.py:89-118 with the same draw order per id: candidate (repeat while interacted), then the interacted pick.
    ``hist_lists`` is mutated like the reference's cache (empty history gets a 0 appended)."""
    ids = np.asarray(ids)
    interacted = np.zeros_like(ids)
    non_interacted = np.zeros_like(ids)
    if mode == 'user':
        cand = list(range(idspace.OI)) + list(range(idspace.target_num_items, idspace.total_num_items))
    else:
        cand = list(range(idspace.OU)) + list(range(idspace.target_num_users, idspace.total_num_users))
    for n, i in enumerate(ids):
        h = hist_lists[i]
        if len(h) == 0:
            h.append(0)
        c = rng.choice(cand, size=1)[0]
        while c in h:
            c = rng.choice(cand, size=1)[0]
        interacted[n] = rng.choice(h, size=1)[0]
        non_interacted[n] = c
    return torch.from_numpy(interacted), torch.from_numpy(non_interacted)


def map_loss(params, idspace, inter, sampled_pos, sampled_neg, margin, lam):
    idx = inter['overlap'].squeeze(1)
    a, b = ('user', 'item') if idspace.mode == 'overlap_users' else ('item', 'user')
    src = params[f'source_{a}_embedding.weight'][idx]
    tgt = params[f'target_{a}_embedding.weight'][idx]
    loss_s = mse_loss(mapping_layer(params, src), tgt)
    other = params[f'source_{b}_embedding.weight']
    mp = mapping_layer(params, other[sampled_pos])
    mn = mapping_layer(params, other[sampled_neg])
    loss_u = triplet_margin_loss(embedding_normalize(tgt), embedding_normalize(mp), embedding_normalize(mn), margin)
    return loss_s + lam * loss_u


def calculate_loss(params, idspace, inter, phase, margin, lam, sampled_pos=None, sampled_neg=None):
    if phase == 'SOURCE':
        return domain_loss(params, inter, 'source', margin)
    if phase == 'OVERLAP':
        return map_loss(params, idspace, inter, sampled_pos, sampled_neg, margin, lam)
    return domain_loss(params, inter, 'target', margin)


def _mapped(params, idspace, ids_, kind):
    n = idspace.OU if kind == 'user' else idspace.OI
    D = params[f'source_{kind}_embedding.weight'].shape[1]
    rep = ids_.repeat(D, 1).transpose(0, 1)
    return torch.where(rep < n, mapping_layer(params, params[f'source_{kind}_embedding.weight'][ids_]),
                       params[f'target_{kind}_embedding.weight'][ids_])


def predict(params, idspace, inter, phase):
    if phase in ('SOURCE', 'TARGET'):
        d = phase.lower()
        ue = embedding_normalize(params[f'{d}_user_embedding.weight'][inter[f'{d}_user_id']])
        ie = embedding_normalize(params[f'{d}_item_embedding.weight'][inter[f'{d}_item_id']])
        return -embedding_distance(ue, ie)
    user, item = inter['target_user_id'], inter['target_item_id']
    if idspace.mode == 'overlap_users':
        ue = _mapped(params, idspace, user, 'user')
        ie = params['target_item_embedding.weight'][item]
    else:
        ue = params['target_user_embedding.weight'][user]
        ie = _mapped(params, idspace, item, 'item')
    return -embedding_distance(embedding_normalize(ue), embedding_normalize(ie))


def full_sort_predict(params, idspace, inter, phase):
    TI = idspace.target_num_items
    if phase == 'SOURCE':
        ue = embedding_normalize(params['source_user_embedding.weight'][inter['source_user_id']])
        W = params['source_item_embedding.weight']
        all_item = torch.cat([embedding_normalize(W[:idspace.OI]), embedding_normalize(W[TI:])], dim=0)
    elif phase == 'TARGET':
        ue = embedding_normalize(params['target_user_embedding.weight'][inter['target_user_id']])
        all_item = embedding_normalize(params['target_item_embedding.weight'][:TI])
    else:
        user = inter['target_user_id']
        if idspace.mode == 'overlap_users':
            ue = _mapped(params, idspace, user, 'user')
            all_item = params['target_item_embedding.weight'][:TI]
        else:
            ue = params['target_user_embedding.weight'][user]
            ov = mapping_layer(params, params['source_item_embedding.weight'][:idspace.OI])
            all_item = torch.cat([ov, params['target_item_embedding.weight'][idspace.OI:TI]], dim=0)
        ue = embedding_normalize(ue)
        all_item = embedding_normalize(all_item)
    U, N = ue.shape[0], all_item.shape[0]
    dist = -2 * torch.matmul(ue, all_item.permute(1, 0))
    dist = dist + torch.sum(ue ** 2, -1).view(U, 1)
    dist = dist + torch.sum(all_item ** 2, -1).view(1, N)
    return -dist.view(-1)
