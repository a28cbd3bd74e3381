"""CLFM restated (oracle; test infrastructure only).  /root/reference recbole_cdr/model/cross_domain_recommender/clfm.py:
source_forward :74-87, target_forward :89-101, calculate_loss :103-123, predict :125-129, full_sort_predict :131-145.
params: {source,target}_{user,item}_embedding.weight, shared_linear.weight [share, Du], source_only_linear.weight,
target_only_linear.weight [Di - share, Du] (each present only when its width is > 0: clfm.py:54-63)."""
import torch

from .losses import bce_loss, emb_loss


def _factors(params, user_e, domain):
    f = []
    if 'shared_linear.weight' in params:
        f.append(user_e @ params['shared_linear.weight'].t())
    if f'{domain}_only_linear.weight' in params:
        f.append(user_e @ params[f'{domain}_only_linear.weight'].t())
    return torch.cat(f, dim=1)


def forward(params, user, item, domain):
    ue = params[f'{domain}_user_embedding.weight'][user]
    ie = params[f'{domain}_item_embedding.weight'][item]
    return torch.sigmoid(torch.mul(_factors(params, ue, domain), ie).sum(dim=1))


def calculate_loss(params, ids, inter, alpha, reg_weight):
    out = []
    for d in ('source', 'target'):
        u, i, y = inter[f'{d}_user_id'], inter[f'{d}_item_id'], inter[f'{d}_label']
        out.append(bce_loss(forward(params, u, i, d), y)
                   + reg_weight * emb_loss(params[f'{d}_user_embedding.weight'][u], params[f'{d}_item_embedding.weight'][i]))
    return out[0] * alpha + out[1] * (1 - alpha)


def predict(params, ids, inter):
    return forward(params, inter['target_user_id'], inter['target_item_id'], 'target')


def full_sort_predict(params, ids, inter):
    ue = params['target_user_embedding.weight'][inter['target_user_id']]
    all_item = params['target_item_embedding.weight'][:ids.target_num_items]
    return torch.matmul(_factors(params, ue, 'target'), all_item.transpose(0, 1)).view(-1)
