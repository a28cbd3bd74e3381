"""Fused row-wise training step: forward + backward + optimizer for one BPR batch without table-sized gradients.

This is the large-table counterpart of ``loss.backward(); optimizer.step()`` in the reference's loop
(recbole_cdr/trainer/trainer.py:59-73 -> recbole ``Trainer._train_epoch``): dense ``[rows, D]`` gradients and a dense
Adam sweep are O(table) per step and impossible at BASELINE config C5 (72 GB of tables).  Here every step touches
only the batch's rows: see csrc/cdr_step.hip for the kernels and DESIGN.md for the (lazy-Adam) semantic note.
"""
import ctypes

import torch

from . import binding as B_

OPT_SGD, OPT_ADAM = 0, 1


class RowwiseState:
    """Per-table optimizer state for the row-wise Adam (allocated lazily; SGD needs none)."""

    def __init__(self, table, opt):
        self.table = table
        self.exp_avg = torch.zeros_like(table) if opt == OPT_ADAM else None
        self.exp_avg_sq = torch.zeros_like(table) if opt == OPT_ADAM else None


class FusedBPRStep:
    """One object per (user table, item table) pair; buffers are sized for ``max_batch`` triples and reused."""

    def __init__(self, user_table, item_table, max_batch, opt='adam', lr=1e-3, betas=(0.9, 0.999), eps=1e-8,
                 weight_decay=0.0, gamma=1e-10, reg_weight=0.0):
        assert user_table.is_cuda and item_table.is_cuda, 'FusedBPRStep needs ROCm device tensors'
        assert user_table.shape[1] == item_table.shape[1]
        self.U, self.I = user_table, item_table
        self.D = user_table.shape[1]
        self.opt = OPT_ADAM if opt == 'adam' else OPT_SGD
        self.lr, self.betas, self.eps, self.wd = lr, betas, eps, weight_decay
        self.gamma, self.reg_weight = gamma, reg_weight
        self.ustate = RowwiseState(user_table, self.opt)
        self.istate = RowwiseState(item_table, self.opt)
        self.step_count = 0
        dev = user_table.device
        Bm = int(max_batch)
        self.max_batch = Bm
        self.GU = torch.empty(Bm, self.D, device=dev, dtype=torch.float32)
        self.GP = torch.empty(Bm, self.D, device=dev, dtype=torch.float32)
        self.out6 = torch.zeros(12, device=dev, dtype=torch.float32)
        self.ukeys = torch.empty(Bm, device=dev, dtype=torch.int32)
        self.uperm = torch.empty(Bm, device=dev, dtype=torch.int32)
        self.ikeys = torch.empty(2 * Bm, device=dev, dtype=torch.int32)
        self.iperm = torch.empty(2 * Bm, device=dev, dtype=torch.int32)
        need = ctypes.c_size_t(0)
        B_._check(B_.load().cdr_sort_workspace_bytes(2 * Bm, max(user_table.shape[0], item_table.shape[0]),
                                                     ctypes.byref(need)), 'cdr_sort_workspace_bytes')
        self.ws_bytes = int(need.value)
        self.ws = torch.empty(self.ws_bytes, device=dev, dtype=torch.uint8)

    def step(self, uid, pid, nid):
        """uid/pid/nid: int64 device tensors [B].  Returns the device tensor out6 (view; [0] = total loss)."""
        B = uid.numel()
        assert B <= self.max_batch
        self.step_count += 1
        s = B_.stream()
        ctxh = B_.ctx(self.U.device)
        B_.call('cdr_bpr_fwd_grad', ctxh, s, B_.f32(self.U), B_.f32(self.I), self.D, B_.i64(uid), B_.i64(pid),
                B_.i64(nid), B, 0, float(self.gamma), float(self.reg_weight), B_.f32(self.out6), B_.f32(self.GU),
                B_.f32(self.GP), 0)
        B_.call('cdr_sort_ids', ctxh, s, B_.i64(uid), B, None, 0, self.U.shape[0], B_.raw(self.ukeys),
                B_.raw(self.uperm), B_.raw(self.ws), self.ws_bytes)
        self._apply(ctxh, self.ustate, self.ukeys, self.uperm, B, self.GU, B, B, self.out6[4:5])
        B_.call('cdr_sort_ids', ctxh, s, B_.i64(pid), B, B_.i64(nid), B, self.I.shape[0], B_.raw(self.ikeys),
                B_.raw(self.iperm), B_.raw(self.ws), self.ws_bytes)
        self._apply(ctxh, self.istate, self.ikeys, self.iperm, 2 * B, self.GP, B, B, self.out6[5:6])
        return self.out6

    def _apply(self, ctxh, st, keys, perm, n, G, neg_start, reg_limit, coef):
        B_.call('cdr_rowwise_apply', ctxh, B_.stream(), self.opt, B_.f32(st.table), B_.f32(st.exp_avg),
                B_.f32(st.exp_avg_sq), self.D, B_.raw(keys), B_.raw(perm), n, B_.f32(G), neg_start, reg_limit,
                B_.f32(coef), float(self.lr), float(self.betas[0]), float(self.betas[1]), float(self.eps),
                float(self.wd), self.step_count, None)
