"""CoNet restated (oracle; test infrastructure only).  /root/reference recbole_cdr/model/cross_domain_recommender/
conet.py: source_forward :105-142, target_forward :144-181, calculate_loss :183-203, predict :205-220,
full_sort_predict :222-242.

params (reference names): {source,target}_{user,item}_embedding.weight,
  {source,target}_crossunit_linear.<l>.{weight,bias}, crossparas.<l>.weight (no bias),
  {source,target}_outputunit.0.{weight,bias}.
Quirks kept: both towers are evaluated in each forward; the cross term H_l is added only on rows whose
user (item) id < overlapped count, PAD id 0 included; reg = sum_l ||H_l||_F un-weighted; predict returns [B,1];
full_sort_predict returns [U,N] from the target tower WITHOUT cross terms.
"""
import torch
import torch.nn.functional as F

from .losses import bce_loss


def n_layers(params):
    return len([k for k in params if k.startswith('crossparas.')])


def towers(params, ids, user, item):
    """Both towers through every cross unit; returns (source_hidden, target_hidden)."""
    s = torch.cat([params['source_user_embedding.weight'][user], params['source_item_embedding.weight'][item]], dim=1)
    t = torch.cat([params['target_user_embedding.weight'][user], params['target_item_embedding.weight'][item]], dim=1)
    mask = (user < ids.OU) if ids.mode == 'overlap_users' else (item < ids.OI)
    m = mask.unsqueeze(1)
    for l in range(n_layers(params)):
        H = params[f'crossparas.{l}.weight'].t()
        so = F.linear(s, params[f'source_crossunit_linear.{l}.weight'], params[f'source_crossunit_linear.{l}.bias'])
        so = torch.where(m, so + torch.mm(t, H), so)
        so = torch.relu(so)
        to = F.linear(t, params[f'target_crossunit_linear.{l}.weight'], params[f'target_crossunit_linear.{l}.bias'])
        to = torch.where(m, to + torch.mm(s, H), to)
        to = torch.relu(to)
        s, t = so, to
    return s, t


def source_forward(params, ids, user, item):
    s, _ = towers(params, ids, user, item)
    return torch.sigmoid(F.linear(s, params['source_outputunit.0.weight'], params['source_outputunit.0.bias'])).squeeze()


def target_forward(params, ids, user, item):
    _, t = towers(params, ids, user, item)
    return torch.sigmoid(F.linear(t, params['target_outputunit.0.weight'], params['target_outputunit.0.bias'])).squeeze()


def calculate_loss(params, ids, inter):
    p_s = source_forward(params, ids, inter['source_user_id'], inter['source_item_id'])
    p_t = target_forward(params, ids, inter['target_user_id'], inter['target_item_id'])
    loss = bce_loss(p_s, inter['source_label']) + bce_loss(p_t, inter['target_label'])
    reg = 0
    for l in range(n_layers(params)):
        reg = reg + torch.norm(params[f'crossparas.{l}.weight'])
    return loss + reg


def _target_tower_plain(params, x):
    for l in range(n_layers(params)):
        x = torch.relu(F.linear(x, params[f'target_crossunit_linear.{l}.weight'], params[f'target_crossunit_linear.{l}.bias']))
    return torch.sigmoid(F.linear(x, params['target_outputunit.0.weight'], params['target_outputunit.0.bias']))


def predict(params, ids, inter):
    x = torch.cat([params['target_user_embedding.weight'][inter['target_user_id']],
                   params['target_item_embedding.weight'][inter['target_item_id']]], dim=1)
    return _target_tower_plain(params, x)                       # [B,1]


def full_sort_predict(params, ids, inter):
    ue = params['target_user_embedding.weight'][inter['target_user_id']]
    all_item = params['target_item_embedding.weight'][:ids.target_num_items]
    N = all_item.shape[0]
    rows = []
    for u in ue:
        x = torch.cat([u.unsqueeze(0).expand(N, -1), all_item], dim=1)
        rows.append(_target_tower_plain(params, x))             # [N,1]
    return torch.cat(rows, dim=1).transpose(0, 1)               # [U,N]
