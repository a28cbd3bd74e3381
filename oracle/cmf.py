"""CMF restated (oracle; test infrastructure only).  /root/reference recbole_cdr/model/cross_domain_recommender/
cmf.py: forward :75-79, calculate_loss :81-99, predict :101-105, full_sort_predict :107-112.
params: user_embedding.weight, item_embedding.weight (shared by both domains)."""
import torch

from .losses import bce_loss, emb_loss


def forward(params, user, item):
    return torch.sigmoid(torch.mul(params['user_embedding.weight'][user], params['item_embedding.weight'][item]).sum(dim=1))


def calculate_loss(params, ids, inter, alpha, lam, gamma):
    U, I = params['user_embedding.weight'], params['item_embedding.weight']
    su, si, sl = inter['source_user_id'], inter['source_item_id'], inter['source_label']
    tu, ti, tl = inter['target_user_id'], inter['target_item_id'], inter['target_label']
    loss_s = bce_loss(forward(params, su, si), sl) + lam * emb_loss(U[su], I[si])
    loss_t = bce_loss(forward(params, tu, ti), tl) + gamma * emb_loss(U[tu], I[ti])
    return loss_s * alpha + loss_t * (1 - alpha)


def predict(params, ids, inter):
    return forward(params, inter['target_user_id'], inter['target_item_id'])


def full_sort_predict(params, ids, inter):
    ue = params['user_embedding.weight'][inter['target_user_id']]
    all_item = params['item_embedding.weight'][:ids.target_num_items]
    return torch.matmul(ue, all_item.transpose(0, 1)).view(-1)
