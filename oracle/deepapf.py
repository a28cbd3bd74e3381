"""DeepAPF restated (oracle; test infrastructure only).  /root/reference recbole_cdr/model/cross_domain_recommender/
deepapf.py: source_forward :69-109, target_forward :111-152, predict :157-161, calculate_loss :163-175.
Quirks kept: the share-branch score is masked where id > overlapped_num (STRICTLY greater: id == overlapped_num, the first
non-overlapped id, keeps its share row; :75,93,117,135); the mask value is -1e31 (softmax weight exactly 0 in fp32);
``self.user_mlp = self.seq = ...; self.item_mlp = self.seq = ...`` (:54-60) registers the item MLP under the name ``seq``
(named_parameters de-duplicates ``item_mlp.*`` away), so the attention MLP's parameters are ``user_mlp.*`` in overlap_users mode
and ``seq.*`` in overlap_items mode."""
import torch

from .losses import bce_loss


def _att(params, pre, x):
    h = torch.relu(x @ params[f'{pre}.0.weight'].t() + params[f'{pre}.0.bias'])
    return h @ params[f'{pre}.2.weight'].t()


def forward(params, ids, user, item, domain):
    if ids.mode == 'overlap_users':
        share = params['share_user_embedding.weight'][user]
        only = params[f'{domain}_user_embedding.weight'][user]
        other = params[f'{domain}_item_embedding.weight'][item]
        mask = (user > ids.overlapped_num_users).unsqueeze(-1)
        pre = 'user_mlp'
    else:
        other = params[f'{domain}_user_embedding.weight'][user]
        share = params['share_item_embedding.weight'][item]
        only = params[f'{domain}_item_embedding.weight'][item]
        mask = (item > ids.overlapped_num_items).unsqueeze(-1)
        pre = 'seq'
    a_share = _att(params, pre, share * other).masked_fill(mask, -1e31)
    a_only = _att(params, pre, only * other)
    alpha = torch.softmax(torch.cat([a_share, a_only], dim=1), dim=1).unsqueeze(1)
    e = (alpha * torch.cat([share.unsqueeze(2), only.unsqueeze(2)], dim=2)).sum(dim=2)
    prod = e * other
    return torch.sigmoid(prod @ params['predict_layer.weight'].t()).squeeze(-1)


def calculate_loss(params, ids, inter):
    ps = forward(params, ids, inter['source_user_id'], inter['source_item_id'], 'source')
    pt = forward(params, ids, inter['target_user_id'], inter['target_item_id'], 'target')
    return bce_loss(ps, inter['source_label']) + bce_loss(pt, inter['target_label'])


def predict(params, ids, inter):
    return forward(params, ids, inter['target_user_id'], inter['target_item_id'], 'target')
