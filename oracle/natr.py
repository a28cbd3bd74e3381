"""NATR restated (oracle; test infrastructure only).  /root/reference recbole_cdr/model/cross_domain_recommender/natr.py:
history info :75-96, phase1 :98-110, phase2_forward :112-156, phase-2 loss :158-168, calculate_loss :170-176, predict :178-191;
set_phase('TARGET') freezes both source tables (:69-73).  recbole RegLoss (un-vendored): sum of the parameters'
2-norms.  The zero fills of :61-65 are overwritten by xavier_normal_initialization (:73)."""
import torch

from .history import history_matrix
from .losses import bce_loss


def history_info(ids, t_pairs, max_inter_length):
    """(history matrix truncated to max_inter_length, lens, mask).  overlap_users: per ITEM its target-domain users;
    overlap_items: per USER their target-domain items.  NOTE the mask is built from the UN-truncated lens (natr.py:80-81)."""
    row = 'item' if ids.mode == 'overlap_users' else 'user'
    mat, _, lens = history_matrix(t_pairs[:, 0], t_pairs[:, 1], ids.total_num_users, ids.total_num_items, row)
    mat = mat[:, :max_inter_length]
    mask = (torch.arange(mat.shape[1]) < lens.unsqueeze(1)).float()
    return mat, lens, mask


def phase1_forward(params, user, item):
    return torch.sigmoid(torch.mul(params['source_user_embedding.weight'][user], params['source_item_embedding.weight'][item]).sum(dim=1))


def phase2_forward(params, ids, hist, user, item):
    mat, _, mask_mat = hist
    user_e = params['target_user_embedding.weight'][user]
    item_e = params['target_item_embedding.weight'][item]
    if ids.mode == 'overlap_items':
        key, src, pu, qi = user, params['source_item_embedding.weight'], user_e, item_e
    else:
        key, src, pu, qi = item, params['source_user_embedding.weight'], item_e, user_e
    bias_mask = torch.where(mask_mat[key].bool(), 0., -10000.0)
    he = src[mat[key]] @ params['transfer_layer.weight'].t() + params['transfer_layer.bias']          # [B, n_hist, Dt]
    att = pu.unsqueeze(1).expand_as(he) * he
    att = (torch.relu(att) @ params['unit_attention_layer.weight'].t() + params['unit_attention_layer.bias']).squeeze(2)
    att = torch.softmax(att + bias_mask, dim=1).unsqueeze(1)
    su = torch.bmm(att, he).squeeze(1)
    dom = lambda x: torch.relu(x) @ params['domain_attention_layer.weight'].t() + params['domain_attention_layer.bias']
    b_s, b_p = dom(su * qi), dom(pu * qi)
    beta_s = torch.exp(b_s) / (torch.exp(b_s) + torch.exp(b_p))
    zu = beta_s * su + (1 - beta_s) * pu
    return torch.sigmoid(torch.mul(zu, qi).sum(dim=1))


def calculate_loss(params, ids, hist, inter, phase, reg_weight):
    if phase == 'SOURCE':
        return bce_loss(phase1_forward(params, inter['source_user_id'], inter['source_item_id']), inter['source_label'])
    if phase == 'TARGET':
        score = phase2_forward(params, ids, hist, inter['target_user_id'], inter['target_item_id'])
        reg = None
        for n in ('target_user_embedding.weight', 'target_item_embedding.weight', 'transfer_layer.weight',
                  'unit_attention_layer.weight', 'domain_attention_layer.weight'):
            reg = params[n].norm(2) if reg is None else reg + params[n].norm(2)
        return bce_loss(score, inter['target_label']) + reg_weight * reg
    return None


def predict(params, ids, hist, inter, phase):
    if phase == 'SOURCE':
        return phase1_forward(params, inter['source_user_id'], inter['source_item_id'])
    return phase2_forward(params, ids, hist, inter['target_user_id'], inter['target_item_id'])
