"""DCDCSR restated (oracle; test infrastructure only).  /root/reference recbole_cdr/model/cross_domain_recommender/dcdcsr.py:
build_unit2pop :83-88, set_phase :90-110, calculate_rec_loss :112-127 (BPR, no regulariser), build_unit_benchmark_embedding
:129-154, maxmin_normalize :167-172, calculate_unit_map_loss :174-182, calculate_loss :192-213, full_sort_predict :215-245,
predict :247-280.  Mapping MLP = recbole MLPLayers([D] + hidden + [D], activation='tanh') -> tanh after EVERY layer, the last
one included (parameters ``mapping_mlp_layers.mlp_layers.{1,4,...}``).
State across phases: ``stage`` is one of 'SOURCE', 'TARGET' (first visits), 'BOTH', 'TARGET2' (second TARGET visit)."""
import numpy as np
import torch

from .history import history_matrix
from .losses import bpr_loss, mse_loss


def unit_pops(ids, s_pairs, t_pairs):
    """(source_pop, target_pop) float [total units]: history lens of the overlapped unit kind (dcdcsr.py:57-62,83-88)."""
    row = 'user' if ids.mode == 'overlap_users' else 'item'
    s = history_matrix(s_pairs[:, 0], s_pairs[:, 1], ids.total_num_users, ids.total_num_items, row)[2].float()
    t = history_matrix(t_pairs[:, 0], t_pairs[:, 1], ids.total_num_users, ids.total_num_items, row)[2].float()
    return s, t


def _mlp(params, x):
    n = 1
    while f'mapping_mlp_layers.mlp_layers.{n}.weight' in params:
        x = torch.tanh(x @ params[f'mapping_mlp_layers.mlp_layers.{n}.weight'].t() + params[f'mapping_mlp_layers.mlp_layers.{n}.bias'])
        n += 3
    return x


def maxmin_normalize(w):
    min_ = torch.amin(w, dim=1, keepdim=True)
    max_ = torch.amax(w, dim=1, keepdim=True)
    mean_ = (max_ + min_) / 2
    return (w - mean_) / (max_ - mean_), mean_, max_


def _unit(ids):
    return ('user', ids.total_num_users, ids.overlapped_num_users, ids.target_num_users) if ids.mode == 'overlap_users' \
        else ('item', ids.total_num_items, ids.overlapped_num_items, ids.target_num_items)


@torch.no_grad()
def build_benchmark_embedding(params, ids, pops, k):
    unit, total, n_over, _ = _unit(ids)
    s_pop, t_pop = pops
    src = params[f'source_{unit}_embedding.weight'][:n_over]
    tgt = params[f'target_{unit}_embedding.weight']
    bench = torch.empty(total, tgt.shape[1])
    for idx in range(n_over):
        den = s_pop[idx] + t_pop[idx]
        if den == 0:
            den = 1
        a_s = s_pop[idx] / den
        bench[idx] = a_s * tgt[idx] + (1 - a_s) * src[idx]
    for idx in range(n_over, total):
        sim_i = torch.mm(src, tgt[idx].unsqueeze(1)).squeeze(1)
        sim, index = torch.topk(sim_i, k=k, dim=0)
        sn = torch.mean(s_pop[index])
        beta = sn / (sn + t_pop[idx])
        sim_e = torch.mm(sim.unsqueeze(0), src[index]).squeeze(0)
        sum_sim = torch.sum(sim) if torch.sum(sim) > 0 else 1
        bench[idx] = (1 - beta) * tgt[idx] + beta * (sim_e / sum_sim)
    return bench


@torch.no_grad()
def build_affine_embedding(params, ids):
    unit, _, _, n_tgt = _unit(ids)
    e, mean_, max_ = maxmin_normalize(params[f'target_{unit}_embedding.weight'][:n_tgt])
    return _mlp(params, e) * (max_ - mean_) + mean_


def map_loss(params, ids, bench, sampled_index):
    unit, _, _, _ = _unit(ids)
    e, _, _ = maxmin_normalize(params[f'target_{unit}_embedding.weight'][sampled_index])
    b, _, _ = maxmin_normalize(bench[sampled_index])
    return mse_loss(_mlp(params, e), b)


def _tables(params, ids, stage, affine):
    if stage == 'SOURCE':
        return params['source_user_embedding.weight'], params['source_item_embedding.weight'], 'source'
    if stage == 'TARGET':
        return params['target_user_embedding.weight'], params['target_item_embedding.weight'], 'target'
    if ids.mode == 'overlap_users':
        return affine, params['target_item_embedding.weight'], 'target'
    return params['target_user_embedding.weight'], affine, 'target'


def rec_loss(params, ids, inter, stage, affine=None):
    U, I, d = _tables(params, ids, stage, affine)
    ue, ie, ne = U[inter[f'{d}_user_id']], I[inter[f'{d}_item_id']], I[inter[f'neg_{d}_item_id']]
    return bpr_loss(torch.mul(ue, ie).sum(dim=1), torch.mul(ue, ne).sum(dim=1))


def predict(params, ids, inter, stage, affine=None):
    U, I, d = _tables(params, ids, stage, affine)
    return torch.mul(U[inter[f'{d}_user_id']], I[inter[f'{d}_item_id']]).sum(dim=1)


def full_sort_predict(params, ids, inter, stage, affine=None):
    U, I, d = _tables(params, ids, stage, affine)
    ue = U[inter[f'{d}_user_id']]
    if stage == 'SOURCE':
        all_item = torch.cat([I[:ids.overlapped_num_items], I[ids.target_num_items:]], dim=0)
    elif stage == 'TARGET' or ids.mode == 'overlap_users':
        all_item = I[:ids.target_num_items]
    else:
        all_item = I                                   # the affine table already has target_num_items rows
    return torch.matmul(ue, all_item.transpose(0, 1))
