"""BiTGCF restated (oracle; test infrastructure only).  /root/reference recbole_cdr/model/cross_domain_recommender/
bitgcf.py: get_norm_adj_mat :92-116, graph_layer :130-135, transfer_layer :137-172, forward :174-205,
calculate_loss :207-250, predict :252-262, full_sort_predict :264-272.

params: {source,target}_{user,item}_embedding.weight.  ``graph`` (see build_graph) carries the two normalised
adjacencies and the four degree vectors.  Dropout is identity unless explicit masks are passed (``forward(masks=...)``): with p>0 the reference
draws from torch's global generator, which is not reproducible outside its own process.
"""
import numpy as np
import torch
import torch.nn.functional as F

from .losses import bce_loss, emb_loss


def norm_adj(pairs, n_users, n_items):
    """D^-1/2 A D^-1/2 of the bipartite graph; degrees = count of distinct neighbours + 1e-7, product formed in
    float64 as ((d_i^-1/2 * a_ij) * d_j^-1/2) then rounded to fp32 (bitgcf.py:103-115).  Returns coalesced COO."""
    pairs = np.unique(np.asarray(pairs, dtype=np.int64), axis=0)
    u, i = pairs[:, 0], pairs[:, 1] + n_users
    row = np.concatenate([u, i])
    col = np.concatenate([i, u])
    n = n_users + n_items
    deg = np.bincount(row, minlength=n).astype(np.float64) + 1e-7
    dinv = np.power(deg, -0.5)
    val = ((dinv[row] * np.float64(1.0)) * dinv[col]).astype(np.float32)
    idx = torch.from_numpy(np.stack([row, col]))
    return torch.sparse_coo_tensor(idx, torch.from_numpy(val), (n, n)).coalesce()


def build_graph(s_pairs, t_pairs, n_users, n_items):
    def deg(pairs, axis, n):
        return torch.from_numpy(np.bincount(np.asarray(pairs)[:, axis], minlength=n).astype(np.float32)).view(-1, 1)
    return {
        'source_adj': norm_adj(s_pairs, n_users, n_items), 'target_adj': norm_adj(t_pairs, n_users, n_items),
        'source_user_deg': deg(s_pairs, 0, n_users), 'target_user_deg': deg(t_pairs, 0, n_users),
        'source_item_deg': deg(s_pairs, 1, n_items), 'target_item_deg': deg(t_pairs, 1, n_items),
    }


def graph_layer(adj, E):
    side = torch.sparse.mm(adj, E)
    new = side + torch.mul(E, side)
    return E + new


def transfer_layer(ids, graph, S, T, lam_s, lam_t):
    nu, ni = ids.total_num_users, ids.total_num_items
    su, si = torch.split(S, [nu, ni])
    tu, ti = torch.split(T, [nu, ni])
    su_lam = lam_s * su + (1 - lam_s) * tu
    tu_lam = lam_t * tu + (1 - lam_t) * su
    si_lam = lam_s * si + (1 - lam_s) * ti
    ti_lam = lam_t * ti + (1 - lam_t) * si
    ul = graph['source_user_deg'] + graph['target_user_deg'] + 1e-7
    u_lap = (graph['source_user_deg'] * su + graph['target_user_deg'] * tu) / ul
    il = graph['source_item_deg'] + graph['target_item_deg'] + 1e-7
    i_lap = (graph['source_item_deg'] * si + graph['target_item_deg'] * ti) / il
    OU, OI = ids.OU, ids.OI
    s_u = torch.cat([(su_lam[:OU] + u_lap[:OU]) / 2, su[OU:]], dim=0)
    t_u = torch.cat([(tu_lam[:OU] + u_lap[:OU]) / 2, tu[OU:]], dim=0)
    s_i = torch.cat([(si_lam[:OI] + i_lap[:OI]) / 2, si[OI:]], dim=0)
    t_i = torch.cat([(ti_lam[:OI] + i_lap[:OI]) / 2, ti[OI:]], dim=0)
    return torch.cat([s_u, s_i], dim=0), torch.cat([t_u, t_i], dim=0)


def forward(params, ids, graph, n_layers, lam_s, lam_t, connect_way, masks=None):
    """``masks`` (optional): {(layer, 's' | 't'): [n_users + n_items, D] tensor of 0 or 1 / (1 - p)} -- nn.Dropout(p) of the graph
    layer's output in training mode (bitgcf.py:66,134) with the mask given explicitly (the reference draws it from torch's global
    generator; a test hands over the mask the product drew so that values, not only statistics, can be compared)."""
    S = torch.cat([params['source_user_embedding.weight'], params['source_item_embedding.weight']], dim=0)
    T = torch.cat([params['target_user_embedding.weight'], params['target_item_embedding.weight']], dim=0)
    s_list, t_list = [S], [T]
    for l in range(n_layers):
        S = graph_layer(graph['source_adj'], S)
        T = graph_layer(graph['target_adj'], T)
        if masks is not None:
            S, T = S * masks[(l, 's')], T * masks[(l, 't')]
        S, T = transfer_layer(ids, graph, S, T, lam_s, lam_t)
        s_list.append(F.normalize(S, p=2, dim=1))      # normalised copies are stacked, raw ones continue (Q10)
        t_list.append(F.normalize(T, p=2, dim=1))
    if connect_way == 'concat':
        Sa, Ta = torch.cat(s_list, 1), torch.cat(t_list, 1)
    else:
        Sa, Ta = torch.mean(torch.stack(s_list, dim=1), dim=1), torch.mean(torch.stack(t_list, dim=1), dim=1)
    nu, ni = ids.total_num_users, ids.total_num_items
    su, si = torch.split(Sa, [nu, ni])
    tu, ti = torch.split(Ta, [nu, ni])
    return su, si, tu, ti


def calculate_loss(params, ids, graph, inter, n_layers, lam_s, lam_t, connect_way, reg_weight, masks=None):
    su_all, si_all, tu_all, ti_all = forward(params, ids, graph, n_layers, lam_s, lam_t, connect_way, masks)
    out = []
    for d, ua, ia in (('source', su_all, si_all), ('target', tu_all, ti_all)):
        u, i, y = inter[f'{d}_user_id'], inter[f'{d}_item_id'], inter[f'{d}_label']
        p = torch.sigmoid(torch.mul(ua[u], ia[i]).sum(dim=1))
        reg = emb_loss(params[f'{d}_user_embedding.weight'][u], params[f'{d}_item_embedding.weight'][i])
        out.append(bce_loss(p, y) + reg_weight * reg)
    return tuple(out)


def predict(params, ids, graph, inter, n_layers, lam_s, lam_t, connect_way):
    _, _, tu, ti = forward(params, ids, graph, n_layers, lam_s, lam_t, connect_way)
    return torch.mul(tu[inter['target_user_id']], ti[inter['target_item_id']]).sum(dim=1)


def full_sort_predict(params, ids, graph, inter, n_layers, lam_s, lam_t, connect_way):
    _, _, tu, ti = forward(params, ids, graph, n_layers, lam_s, lam_t, connect_way)
    return torch.matmul(tu[inter['target_user_id']], ti[:ids.target_num_items].transpose(0, 1)).view(-1)
