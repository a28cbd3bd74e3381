"""Row-sharded tables + the all-to-all exchange of one BPR training step (one process per GPU, RCCL over xGMI).

The reference is single-device (no distributed code at all); this module is the new multi-GPU path of
north_star.  Parity is defined against the single-device result: with the same GLOBAL batch, the summed per-rank loss
partials and every touched row after the step equal the 1-GPU fused step (tests/test_shard_gloo.py, world_size 2).

Layout    row r of a table lives on rank r % G at local row r // G (balanced for any id distribution that is not
          adversarially strided; contiguous id blocks of one domain spread over all ranks).
Batch     data-parallel: every rank brings its own B triples (global ids).
Exchange  per table: ids bucketed by owner -> all_to_all(ids) -> owners gather rows -> all_to_all(rows) ->
          fused forward/compact-gradient kernel on the received rows -> all_reduce of the three loss sums (the EmbLoss
          norms are over the GLOBAL batch) -> per-occurrence gradient rows -> all_to_all back -> owners sort + row-wise
          optimizer.  Only `torch.distributed` collectives move data; bucketing is index plumbing.

The arithmetic is behind ``self.ops`` so that the CPU (gloo) tests can drive the same exchange code with stand-in
compute; the product default (``NativeOps``) is libcdrhip only.
"""
import ctypes

import torch
import torch.distributed as dist

OPT_SGD, OPT_ADAM = 0, 1


def shard_rows(total_rows, world, rank):
    """Number of rows r in [0, total_rows) with r % world == rank."""
    return (total_rows - rank + world - 1) // world


def shard_of(full_table, world, rank):
    """The rows of a full table that rank owns, in local order (test helper / checkpoint loading)."""
    return full_table[rank::world].contiguous()


class NativeOps:
    """libcdrhip kernels (csrc/cdr_rows.hip, csrc/cdr_step.hip)."""

    def __init__(self, device):
        from . import binding as B_
        self.B_ = B_
        self.device = device
        self._ws = None

    def gather_rows(self, table, local_ids):
        B_ = self.B_
        out = torch.empty(local_ids.numel(), table.shape[1], device=table.device, dtype=torch.float32)
        if local_ids.numel():
            B_.call('cdr_gather_rows', B_.stream(), B_.f32(table), table.shape[1], B_.i64(local_ids), local_ids.numel(),
                    B_.f32(out))
        return out

    def fwd_grad(self, urows, irows, upos, ppos, npos, B_mean, gamma, reg_weight, out, GU, GP):
        B_ = self.B_
        B_.call('cdr_bpr_fwd_grad', B_.ctx(self.device), B_.stream(), B_.f32(urows), B_.f32(irows), urows.shape[1],
                B_.i64(upos), B_.i64(ppos), B_.i64(npos), upos.numel(), int(B_mean), float(gamma), float(reg_weight),
                B_.f32(out), B_.f32(GU), B_.f32(GP))

    def finish_sums(self, sums3, B_mean, reg_weight, out):
        B_ = self.B_
        B_.call('cdr_loss_finish_sums', B_.stream(), B_.f32(sums3), int(B_mean), float(reg_weight), B_.f32(out))

    def build_grad_rows(self, G, order, neg_start, reg_limit, rows, coef):
        B_ = self.B_
        out = torch.empty(order.numel(), G.shape[1], device=G.device, dtype=torch.float32)
        B_.call('cdr_build_grad_rows', B_.stream(), B_.f32(G), B_.i64(order), order.numel(), G.shape[1], int(neg_start),
                int(reg_limit), B_.f32(rows), B_.f32(coef), B_.f32(out))
        return out

    def sort_apply(self, table, state, local_ids, grads, opt, hp, step):
        """Owner side: segment the received (local row, gradient row) pairs and apply the optimizer in place."""
        B_ = self.B_
        n = local_ids.numel()
        if n == 0:
            return
        need = ctypes.c_size_t(0)
        B_._check(B_.load().cdr_sort_workspace_bytes(n, table.shape[0], ctypes.byref(need)), 'cdr_sort_workspace_bytes')
        if self._ws is None or self._ws.numel() < need.value:
            self._ws = torch.empty(int(need.value), device=table.device, dtype=torch.uint8)
        keys = torch.empty(n, device=table.device, dtype=torch.int32)
        perm = torch.empty(n, device=table.device, dtype=torch.int32)
        ctxh = B_.ctx(self.device)
        B_.call('cdr_sort_ids', ctxh, B_.stream(), B_.i64(local_ids), n, None, 0, table.shape[0], B_.raw(keys),
                B_.raw(perm), B_.raw(self._ws), self._ws.numel())
        m, v = (state if state is not None else (None, None))
        B_.call('cdr_rowwise_apply', ctxh, B_.stream(), opt, B_.f32(table), B_.f32(m), B_.f32(v), table.shape[1],
                B_.raw(keys), B_.raw(perm), n, B_.f32(grads), n, 0, None, float(hp['lr']), float(hp['b1']),
                float(hp['b2']), float(hp['eps']), float(hp['wd']), int(step))


class Route:
    """Bucketing of one id list by owner rank (index plumbing only)."""

    def __init__(self, ids, world):
        owner = ids % world
        self.order = torch.argsort(owner, stable=True)           # occurrence index, grouped by owner
        self.send_counts = torch.bincount(owner, minlength=world)
        self.local_sorted = (ids // world)[self.order].contiguous()
        self.pos = torch.empty_like(self.order)
        self.pos[self.order] = torch.arange(ids.numel(), device=ids.device)


def _a2a(out_numel_list, inp, in_splits, out_splits, group, trailing=()):
    out = torch.empty((sum(out_splits),) + tuple(trailing), device=inp.device, dtype=inp.dtype)
    dist.all_to_all_single(out, inp, output_split_sizes=out_splits, input_split_sizes=in_splits, group=group)
    return out


class ShardedBPRStep:
    """Multi-GPU counterpart of fused.FusedBPRStep.  ``user_shard`` / ``item_shard`` are this rank's rows
    (row r % G == rank, local index r // G)."""

    def __init__(self, user_shard, item_shard, n_users_total, n_items_total, max_batch, opt='adam', lr=1e-3,
                 betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, gamma=1e-10, reg_weight=0.0, group=None, ops=None):
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        assert user_shard.shape[0] == shard_rows(n_users_total, self.world, self.rank)
        assert item_shard.shape[0] == shard_rows(n_items_total, self.world, self.rank)
        self.U, self.I = user_shard, item_shard
        self.D = user_shard.shape[1]
        self.opt = OPT_ADAM if opt == 'adam' else OPT_SGD
        self.hp = {'lr': lr, 'b1': betas[0], 'b2': betas[1], 'eps': eps, 'wd': weight_decay}
        self.gamma, self.reg_weight = gamma, reg_weight
        dev = user_shard.device
        self.ops = ops if ops is not None else NativeOps(dev)
        self.ustate = (torch.zeros_like(user_shard), torch.zeros_like(user_shard)) if self.opt == OPT_ADAM else None
        self.istate = (torch.zeros_like(item_shard), torch.zeros_like(item_shard)) if self.opt == OPT_ADAM else None
        self.max_batch = int(max_batch)
        self.GU = torch.empty(self.max_batch, self.D, device=dev, dtype=torch.float32)
        self.GP = torch.empty(self.max_batch, self.D, device=dev, dtype=torch.float32)
        self.out = torch.zeros(12, device=dev, dtype=torch.float32)
        self.step_count = 0

    def loss_value(self):
        return self.out[0]

    def step(self, uid, pid, nid):
        G, grp = self.world, self.group
        B = uid.numel()
        assert B <= self.max_batch
        self.step_count += 1
        dev = uid.device

        # ---- 1. route ids to the rows' owners ----------------------------------------------------------------
        ru = Route(uid, G)
        ri = Route(torch.cat([pid, nid]), G)
        counts = torch.cat([ru.send_counts, ri.send_counts, torch.tensor([B], device=dev, dtype=torch.int64)])
        gathered = [torch.empty_like(counts) for _ in range(G)]
        dist.all_gather(gathered, counts, group=grp)
        all_counts = torch.stack(gathered).tolist()                      # one host sync per step
        u_send = [int(c) for c in all_counts[self.rank][:G]]
        i_send = [int(c) for c in all_counts[self.rank][G:2 * G]]
        u_recv = [int(all_counts[r][self.rank]) for r in range(G)]
        i_recv = [int(all_counts[r][G + self.rank]) for r in range(G)]
        B_global = sum(int(all_counts[r][2 * G]) for r in range(G))

        u_req = _a2a(None, ru.local_sorted, u_send, u_recv, grp)          # local rows other ranks want from me
        i_req = _a2a(None, ri.local_sorted, i_send, i_recv, grp)

        # ---- 2. owners gather, rows travel back -------------------------------------------------------------
        urows = _a2a(None, self.ops.gather_rows(self.U, u_req), u_recv, u_send, grp, (self.D,))
        irows = _a2a(None, self.ops.gather_rows(self.I, i_req), i_recv, i_send, grp, (self.D,))

        # ---- 3. fused forward + compact gradients on the received rows (positions instead of ids) ------------
        self.ops.fwd_grad(urows, irows, ru.pos, ri.pos[:B].contiguous(), ri.pos[B:].contiguous(), B_global, self.gamma,
                          self.reg_weight, self.out, self.GU, self.GP)
        sums = self.out[6:9].clone()
        dist.all_reduce(sums, group=grp)                                   # loss mean and EmbLoss norms are GLOBAL
        self.ops.finish_sums(sums, B_global, self.reg_weight, self.out)

        # ---- 4. per-occurrence gradient rows (owner order), EmbLoss term folded in, back to the owners ---------
        gu = self.ops.build_grad_rows(self.GU, ru.order, B, B, urows, self.out[4:5])
        gi = self.ops.build_grad_rows(self.GP, ri.order, B, B, irows, self.out[5:6])
        gu_recv = _a2a(None, gu, u_send, u_recv, grp, (self.D,))
        gi_recv = _a2a(None, gi, i_send, i_recv, grp, (self.D,))

        # ---- 5. owners: segment by local row, row-wise optimizer ----------------------------------------------
        self.ops.sort_apply(self.U, self.ustate, u_req, gu_recv, self.opt, self.hp, self.step_count)
        self.ops.sort_apply(self.I, self.istate, i_req, gi_recv, self.opt, self.hp, self.step_count)
        return self.out
