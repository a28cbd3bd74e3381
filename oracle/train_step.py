"""CPU restatement of one training step (oracle; test infrastructure + bench.py's cpu_baseline leg only).

Two flavours of "what the reference's loop does per batch" (recbole_cdr/trainer/trainer.py:59-73 -> recbole
Trainer._train_epoch: zero_grad -> calculate_loss -> backward -> optimizer.step):

  dense_step    the reference's literal semantics: autograd into dense [rows, D] table gradients + torch.optim.Adam over
                every parameter (recbole builds Adam(model.parameters()), properties/overall.yaml:20-21).  O(table).
  rowwise_step  the same loss and the same per-row gradients, applied to the touched rows only (lazy Adam / SGD) -- the
                algorithm the GPU fused step implements, and the only one that can run at BASELINE config C5 sizes.
"""
import torch

from . import emcdr
from .losses import bpr_loss, emb_loss


def dense_step(params, ids, inter, phase, optimizer, latent_factor_model='BPR', reg_weight=0.01):
    optimizer.zero_grad()
    loss = emcdr.calculate_loss(params, ids, inter, phase, latent_factor_model, reg_weight)
    loss.sum().backward()
    optimizer.step()
    return loss.detach()


class RowwiseAdamState:
    def __init__(self, table):
        self.m = torch.zeros_like(table)
        self.v = torch.zeros_like(table)


def _apply_rows(W, state, rows, grad_rows, opt, lr, step, b1=0.9, b2=0.999, eps=1e-8):
    """rows: unique ids, grad_rows: summed gradient per unique id."""
    if opt == 'sgd':
        W[rows] -= lr * grad_rows
        return
    bc1, bc2 = 1 - b1 ** step, 1 - b2 ** step
    m = state.m[rows]
    m = m + (grad_rows - m) * (1 - b1)
    v = b2 * state.v[rows] + (1 - b2) * grad_rows * grad_rows
    state.m[rows], state.v[rows] = m, v
    W[rows] -= (lr / bc1) * (m / (v.sqrt() / (bc2 ** 0.5) + eps))


def rowwise_step(U, I, ustate, istate, uid, pid, nid, step, opt='adam', lr=1e-3, reg_weight=0.01):
    """One BPR step on the touched rows only; same loss/gradients as emcdr.domain_loss (emcdr.py:123-131)."""
    ue = U[uid].requires_grad_(True)
    pe = I[pid].requires_grad_(True)
    ne = I[nid].requires_grad_(True)
    loss = bpr_loss((ue * pe).sum(1), (ue * ne).sum(1)) + reg_weight * emb_loss(ue, pe)
    gu, gp, gn = torch.autograd.grad(loss.sum(), [ue, pe, ne])
    with torch.no_grad():
        ru, inv_u = torch.unique(uid, return_inverse=True)
        GU = torch.zeros(ru.numel(), U.shape[1]).index_add_(0, inv_u, gu)
        items = torch.cat([pid, nid])
        ri, inv_i = torch.unique(items, return_inverse=True)
        GI = torch.zeros(ri.numel(), I.shape[1]).index_add_(0, inv_i, torch.cat([gp, gn]))
        _apply_rows(U, ustate, ru, GU, opt, lr, step)
        _apply_rows(I, istate, ri, GI, opt, lr, step)
    return loss.detach()


def rowwise_map_step(map_params, S, T, sstate, tstate, idx, step_s, step_t, map_optimizer, opt='adam', lr=1e-3):
    """EMCDR OVERLAP-phase step (emcdr.py:133-137 map loss) with the two embedding tables updated on the touched rows
    only; the mapping parameters (``map_params``: the 'mapping.*' leaf tensors, requires_grad) take ``map_optimizer``'s
    dense step, exactly as in the reference."""
    idx = idx.reshape(-1)
    src = S[idx].requires_grad_(True)
    tgt = T[idx].requires_grad_(True)
    map_optimizer.zero_grad()
    loss = torch.nn.functional.mse_loss(emcdr.mapping(map_params, src), tgt)
    loss.backward()
    with torch.no_grad():
        rows, inv = torch.unique(idx, return_inverse=True)
        GS = torch.zeros(rows.numel(), S.shape[1]).index_add_(0, inv, src.grad)
        GT = torch.zeros(rows.numel(), T.shape[1]).index_add_(0, inv, tgt.grad)
        _apply_rows(S, sstate, rows, GS, opt, lr, step_s)
        _apply_rows(T, tstate, rows, GT, opt, lr, step_t)
    map_optimizer.step()
    return loss.detach()


def rowwise_point_step(U, I, ustate, istate, uid, iid, label, step_u, step_i, opt='adam', lr=1e-3, reg_weight=0.01, loss='mse'):
    """Pointwise step on the touched rows only; loss as emcdr.domain_loss for MF (emcdr.py:111-122) or BCE on sigmoid(dot)
    (cmf.py:75-99)."""
    from .losses import mse_loss
    ue = U[uid].requires_grad_(True)
    ie = I[iid].requires_grad_(True)
    dot = (ue * ie).sum(1)
    main = mse_loss(dot, label) if loss == 'mse' else torch.nn.functional.binary_cross_entropy(torch.sigmoid(dot), label)
    total = main + reg_weight * emb_loss(ue, ie)
    gu, gi = torch.autograd.grad(total.sum(), [ue, ie])
    with torch.no_grad():
        ru, inv_u = torch.unique(uid, return_inverse=True)
        ri, inv_i = torch.unique(iid, return_inverse=True)
        _apply_rows(U, ustate, ru, torch.zeros(ru.numel(), U.shape[1]).index_add_(0, inv_u, gu), opt, lr, step_u)
        _apply_rows(I, istate, ri, torch.zeros(ri.numel(), I.shape[1]).index_add_(0, inv_i, gi), opt, lr, step_i)
    return total.detach()
