"""Row-sharded tables + the exchange of one BPR training step over the GPUs of a node (one process per GPU, RCCL/xGMI).

The reference is single-device (no distributed code at all); this module is the new multi-GPU path of
north_star.  Parity is defined against the single-device result: with the same GLOBAL batch, the loss and every
touched row after the step equal the 1-GPU fused step (tests/test_shard_gloo.py at world_size 2 on CPU/gloo;
tests/test_gpu_parity.py::test_sharded_* over a 1-rank RCCL group on the GPU).

Layout    row r of a table lives on rank r % G at local row r // G.
Batch     data-parallel: every rank brings its own B triples (global ids).
Exchange  USER-ALIGNED: each triple first travels (24 B) to the rank that owns its user row, so user rows and user
          gradients never cross xGMI.  Only item rows do:
            0. all_to_all   triples -> owner of u                                  (ids only)
            1. all_to_all   local item ids -> item owners ; owners gather ; all_to_all rows back   (2 rows / triple)
            2. fused forward + compact gradients on (local user shard, received item rows); all_reduce of the three
               loss sums -- the mean and the EmbLoss norms are over the GLOBAL batch
            3. user rows: local sort + row-wise optimizer.  item rows: the forward kernel has already written the
               per-occurrence gradient rows into the send buffer in owner order -> all_to_all back -> owners sort +
               apply, adding the EmbLoss term for the occurrences the requester tagged as positives (bit 62 of the id).
          Collective sizes are exact (bucket counts are all-gathered; one host sync per routing stage).
Overlap   ``step_gen`` is a generator that yields exactly where the host has to wait for bucket counts;
          ``run_pipelined`` round-robins several steps (the SOURCE and TARGET domains touch disjoint tables), each on its
          own HIP stream and process group, so one domain's all-to-alls overlap the other's kernels.

The arithmetic is behind ``self.ops`` so that the CPU (gloo) tests can drive the same exchange code with stand-in
compute; the product default (``NativeOps``) is libcdrhip only.
"""
import contextlib
import ctypes

import torch
import torch.distributed as dist

OPT_SGD, OPT_ADAM = 0, 1
TAG_BIT = 1 << 62                       # marks a positive-item occurrence in the ids a rank sends to an item owner
TAG_MASK = TAG_BIT - 1


def shard_rows(total_rows, world, rank):
    """Number of rows r in [0, total_rows) with r % world == rank."""
    return (total_rows - rank + world - 1) // world


def shard_of(full_table, world, rank):
    """The rows of a full table that rank owns, in local order (test helper / checkpoint loading)."""
    return full_table[rank::world].contiguous()


class NativeOps:
    """libcdrhip kernels (csrc/cdr_route.hip, cdr_rows.hip, cdr_step.hip)."""

    def __init__(self, device):
        from . import binding as B_
        self.B_ = B_
        self.device = device
        self._ws = {}

    def _workspace(self, key, nbytes, device):
        ws = self._ws.get(key)
        if ws is None or ws.numel() < nbytes:
            ws = torch.empty(int(nbytes), device=device, dtype=torch.uint8)
            self._ws[key] = ws
        return ws

    def route(self, ids0, ids1, world):
        """-> (perm uint32-as-int32 [n], counts int64 [world]) : stable bucketing of ids0 ++ ids1 by id % world."""
        B_ = self.B_
        n0 = ids0.numel()
        n1 = ids1.numel() if ids1 is not None else 0
        n = n0 + n1
        dev = ids0.device
        perm = torch.empty(n, device=dev, dtype=torch.int32)
        counts = torch.zeros(world, device=dev, dtype=torch.int64)
        if n == 0:
            return perm, counts
        need = ctypes.c_size_t(0)
        B_._check(B_.load().cdr_route_workspace_bytes(n, world, ctypes.byref(need)), 'cdr_route_workspace_bytes')
        ws = self._workspace(('route', torch.cuda.current_stream().cuda_stream), need.value, dev)
        B_.call('cdr_route_by_owner', B_.ctx(self.device), B_.stream(), B_.i64(ids0), n0, B_.i64(ids1) if n1 else None, n1,
                int(world), B_.raw(perm), B_.i64(counts), B_.raw(ws), ws.numel())
        return perm, counts

    def permute(self, src0, src1, perm, divisor, flag_below=0):
        """out[q] = (src0 ++ src1)[perm[q]] // divisor, bit 62 set where perm[q] < flag_below (positive-item tag)."""
        B_ = self.B_
        n = perm.numel()
        out = torch.empty(n, device=src0.device, dtype=torch.int64)
        if n:
            B_.call('cdr_permute_i64', B_.stream(), B_.i64(src0), src0.numel(), B_.i64(src1) if src1 is not None else None,
                    B_.raw(perm), n, int(divisor), int(flag_below), B_.i64(out))
        return out

    def inverse_perm(self, perm):
        B_ = self.B_
        pos = torch.empty(perm.numel(), device=perm.device, dtype=torch.int64)
        if perm.numel():
            B_.call('cdr_inverse_perm', B_.stream(), B_.raw(perm), perm.numel(), B_.i64(pos))
        return pos

    def gather_rows(self, table, local_ids):
        B_ = self.B_
        out = torch.empty(local_ids.numel(), table.shape[1], device=table.device, dtype=torch.float32)
        if local_ids.numel():
            B_.call('cdr_gather_rows', B_.stream(), B_.f32(table), table.shape[1], B_.i64(local_ids), local_ids.numel(),
                    B_.f32(out))
        return out

    def fwd_grad(self, utab, itab, uidx, pidx, nidx, B_mean, gamma, reg_weight, out, GU, GI, scatter=True):
        """scatter: GI [rows of itab, D], item gradients written at the item rows' own positions (+ at pidx, - at nidx);
        else GI [B, D] = g u per triple (compact)."""
        B_ = self.B_
        B_.call('cdr_bpr_fwd_grad', B_.ctx(self.device), B_.stream(), B_.f32(utab), B_.f32(itab), utab.shape[1],
                B_.i64(uidx), B_.i64(pidx), B_.i64(nidx), uidx.numel(), int(B_mean), float(gamma), float(reg_weight),
                B_.f32(out), B_.f32(GU), B_.f32(GI), 1 if scatter else 0)

    def dedup(self, ids0, ids1, world, local_rows):
        """The distinct item rows behind the occurrences ids0 ++ ids1 (global ids): plan with ``uniq_local`` (local rows,
        grouped by owner, ascending), ``umap`` (occurrence -> unique index), ``counts`` (uniques per owner, device) and the
        sorted occurrence order that ``segsum`` sums over."""
        B_ = self.B_
        dev = ids0.device
        n = ids0.numel() + ids1.numel()
        lb = max(int(local_rows - 1).bit_length(), 1)
        key0 = ((ids0 % world) << lb) | (ids0 // world)                      # index plumbing: owner-major sort key
        key1 = ((ids1 % world) << lb) | (ids1 // world)
        need = ctypes.c_size_t(0)
        B_._check(B_.load().cdr_sort_workspace_bytes(n, world << lb, ctypes.byref(need)), 'cdr_sort_workspace_bytes')
        need2 = ctypes.c_size_t(0)
        B_._check(B_.load().cdr_dedup_workspace_bytes(n, ctypes.byref(need2)), 'cdr_dedup_workspace_bytes')
        ws = self._workspace(('dedup', torch.cuda.current_stream().cuda_stream), max(need.value, need2.value), dev)
        keys = torch.empty(n, device=dev, dtype=torch.int32)
        perm = torch.empty(n, device=dev, dtype=torch.int32)
        ctxh = B_.ctx(self.device)
        B_.call('cdr_sort_ids', ctxh, B_.stream(), B_.i64(key0), ids0.numel(), B_.i64(key1), ids1.numel(), world << lb,
                B_.raw(keys), B_.raw(perm), B_.raw(ws), ws.numel())
        uidx = torch.empty(n, device=dev, dtype=torch.int32)
        uniq_local = torch.empty(n, device=dev, dtype=torch.int64)
        umap = torch.empty(n, device=dev, dtype=torch.int64)
        counts = torch.empty(world + 1, device=dev, dtype=torch.int64)           # [world] counts + one scratch slot
        n_uniq = torch.empty(1, device=dev, dtype=torch.int64)
        B_.call('cdr_dedup_sorted', B_.stream(), B_.raw(keys), B_.raw(perm), n, int(world), lb, B_.raw(uidx), B_.i64(uniq_local),
                B_.i64(umap), B_.i64(counts), B_.i64(n_uniq), B_.raw(ws), ws.numel())
        return {'keys': keys, 'perm': perm, 'uidx': uidx, 'uniq_local': uniq_local, 'umap': umap, 'counts': counts[:world], 'n': n}

    def segsum(self, plan, G_rows, neg_start, rows, reg_coef, n_uniq):
        """One summed gradient row per distinct item (+ the EmbLoss term coef * #positives * row) -> [n_uniq, D]."""
        B_ = self.B_
        D = G_rows.shape[1]
        out = torch.empty(n_uniq, D, device=G_rows.device, dtype=torch.float32)
        if plan['n']:
            B_.call('cdr_segsum_rows', B_.ctx(self.device), B_.stream(), B_.raw(plan['keys']), B_.raw(plan['perm']),
                    B_.raw(plan['uidx']), plan['n'], B_.f32(G_rows), int(neg_start), D, B_.f32(rows), B_.f32(reg_coef), B_.f32(out))
        return out

    def batch_norms(self, utab, irows, u_loc, ip):
        """{0, sum ||U[u]||^2, sum ||irows[ip]||^2} of this rank's triples (device [3]): all-reduced, they give the EmbLoss coefficients of
        the GLOBAL batch before any row is updated."""
        B_ = self.B_
        sums = torch.empty(3, device=utab.device, dtype=torch.float32)
        B_.call('cdr_batch_norm_sums', B_.ctx(self.device), B_.stream(), B_.f32(utab), B_.f32(irows), utab.shape[1], B_.i64(u_loc),
                B_.i64(ip), u_loc.numel(), B_.f32(sums))
        return sums

    def local_step(self, utab, ustate, irows, u_loc, ip, in_, B_mean, gamma, reg_weight, opt, hp, step, out):
        """The requester's half of the row-sharded step on the one-GPU kernels (cdr_bpr_shard_local_step): user rows that occur once are
        updated by the forward pass itself, duplicate user rows by the segmented apply; returns GP [Bl, D] (g_t u_t per triple) for
        ``segsum``.  ``out[4:6]`` = the coefficients (in); ``out[6:9]`` = this rank's loss / norm sums (out)."""
        B_ = self.B_
        Bl, D, dev = u_loc.numel(), utab.shape[1], utab.device
        key = ('local', torch.cuda.current_stream().cuda_stream, utab.data_ptr())
        buf = self._ws.get(key)
        if buf is None or buf['cap'] < Bl:
            cap = max(int(Bl * 1.25), 1024)
            words, need = ctypes.c_int64(0), ctypes.c_size_t(0)
            B_._check(B_.load().cdr_bpr_step_fused_heads_words(cap, ctypes.byref(words)), 'cdr_bpr_step_fused_heads_words')
            B_._check(B_.load().cdr_sort_workspace_bytes(cap, utab.shape[0], ctypes.byref(need)), 'cdr_sort_workspace_bytes')
            buf = self._ws[key] = {
                'cap': cap, 'GU': torch.empty(cap, D, device=dev, dtype=torch.float32), 'GP': torch.empty(cap, D, device=dev, dtype=torch.float32),
                'keys': torch.empty(cap, device=dev, dtype=torch.int32), 'perm': torch.empty(cap, device=dev, dtype=torch.int32),
                'flags': torch.zeros(4 * cap, device=dev, dtype=torch.uint8),        # the item bytes of every group stay 0: no item row is updated here
                'heads': torch.empty(int(words.value), device=dev, dtype=torch.int32),
                'ws': torch.empty(int(need.value), device=dev, dtype=torch.uint8)}
        m, v = ustate if ustate is not None else (None, None)
        B_.call('cdr_bpr_shard_local_step', B_.ctx(self.device), B_.stream(), opt, B_.f32(utab), B_.f32(m), B_.f32(v), utab.shape[0],
                B_.f32(irows), D, B_.i64(u_loc), B_.i64(ip), B_.i64(in_), Bl, int(B_mean), float(gamma), float(reg_weight), float(hp['lr']),
                float(hp['b1']), float(hp['b2']), float(hp['eps']), float(hp['wd']), int(step), B_.f32(out), B_.f32(buf['GU']),
                B_.f32(buf['GP']), B_.raw(buf['keys']), B_.raw(buf['perm']), B_.raw(buf['flags']), B_.raw(buf['heads']), B_.raw(buf['ws']),
                buf['ws'].numel())
        return buf['GP']

    def finish_sums(self, sums3, B_mean, reg_weight, out):
        B_ = self.B_
        B_.call('cdr_loss_finish_sums', B_.stream(), B_.f32(sums3), int(B_mean), float(reg_weight), B_.f32(out))

    def sort_apply(self, table, state, local_ids, grads, opt, hp, step, reg_limit=0, reg_coef=None, tagged=False):
        """Segment (local row, gradient row) pairs by row and apply the optimizer in place.  The EmbLoss term
        ``reg_coef * count * W[r]`` is added inside the kernel: ``count`` = occurrences below ``reg_limit``, or (``tagged``)
        occurrences whose id carries bit 62 (positive items, tagged by the requesting rank)."""
        B_ = self.B_
        n = local_ids.numel()
        if n == 0:
            return
        need = ctypes.c_size_t(0)
        B_._check(B_.load().cdr_sort_workspace_bytes(n, table.shape[0], ctypes.byref(need)), 'cdr_sort_workspace_bytes')
        ws = self._workspace(('sort', torch.cuda.current_stream().cuda_stream), need.value, table.device)
        keys = torch.empty(n, device=table.device, dtype=torch.int32)
        perm = torch.empty(n, device=table.device, dtype=torch.int32)
        ctxh = B_.ctx(self.device)
        B_.call('cdr_sort_ids', ctxh, B_.stream(), B_.i64(local_ids), n, None, 0, table.shape[0], B_.raw(keys),
                B_.raw(perm), B_.raw(ws), ws.numel())
        m, v = (state if state is not None else (None, None))
        B_.call('cdr_rowwise_apply', ctxh, B_.stream(), opt, B_.f32(table), B_.f32(m), B_.f32(v), table.shape[1],
                B_.raw(keys), B_.raw(perm), n, B_.f32(grads), n, int(reg_limit), B_.f32(reg_coef), float(hp['lr']),
                float(hp['b1']), float(hp['b2']), float(hp['eps']), float(hp['wd']), int(step),
                B_.i64(local_ids) if tagged else None, 0)


    # ---- round 6: the step on its own passes (cdr_route_triples / cdr_bpr_shard_plan / cdr_gather_rows_norms / cdr_shard_norm_sums /
    # cdr_bpr_shard_step / cdr_shard_owner_apply) -------------------------------------------------------------------------------------
    def route_triples(self, uid, pid, nid, world):
        """-> (send3 int64 [B, 3] = {uid // world, pid, nid} in owner order of uid % world (stable), counts int64 [world])."""
        B_ = self.B_
        n, dev = uid.numel(), uid.device
        send3 = torch.empty(n, 3, device=dev, dtype=torch.int64)
        counts = torch.zeros(world, device=dev, dtype=torch.int64)
        if n == 0:
            return send3, counts
        need = ctypes.c_size_t(0)
        B_._check(B_.load().cdr_route_triples_workspace_bytes(n, world, ctypes.byref(need)), 'cdr_route_triples_workspace_bytes')
        ws = self._workspace(('route3', torch.cuda.current_stream().cuda_stream), need.value, dev)
        B_.call('cdr_route_triples', B_.ctx(self.device), B_.stream(), B_.i64(uid), B_.i64(pid), B_.i64(nid), n, int(world), B_.i64(send3),
                B_.i64(counts), B_.raw(ws), ws.numel())
        return send3, counts

    def plan(self, recv3, world, user_rows, item_local_rows, D, slot=0):
        """One sort for both lists of the routed triples -> the flags / duplicate heads of the forward-and-update pass and the request list
        (``uniq_local`` grouped by owner, ``umap`` occurrence -> slot, ``counts`` per owner).  Buffers persist per (stream, table, ``slot``): a
        prefetching caller alternates two slots, the next step's plan is built while this step's kernels still read theirs."""
        B_ = self.B_
        Bl, dev = recv3.shape[0], recv3.device
        key = ('plan', torch.cuda.current_stream().cuda_stream, int(user_rows), int(item_local_rows), int(slot))
        buf = self._ws.get(key)
        if buf is None or buf['cap'] < Bl:
            if buf is not None:
                self._ws.setdefault('retired', []).append(buf)      # a main stage on another stream may still read the smaller set: never hand it back
            cap = max(int(Bl * 1.25), 1024)
            words, need = ctypes.c_int64(0), ctypes.c_size_t(0)
            B_._check(B_.load().cdr_bpr_shard_plan_sizes(cap, int(user_rows), int(item_local_rows), int(world), ctypes.byref(words), ctypes.byref(need)),
                      'cdr_bpr_shard_plan_sizes')
            e = lambda n, dt: torch.empty(n, device=dev, dtype=dt)  # noqa: E731
            buf = self._ws[key] = {
                'cap': cap, 'u_loc': e(cap, torch.int64), 'keys': e(3 * cap, torch.int32), 'perm': e(3 * cap, torch.int32),
                'flags': torch.zeros(4 * cap, device=dev, dtype=torch.uint8), 'heads': e(int(words.value), torch.int32),
                'uidx': e(2 * cap, torch.int32), 'uniq_local': e(2 * cap, torch.int64), 'umap': e(2 * cap, torch.int64),
                'counts': e(world + 1, torch.int64), 'n_uniq': e(1, torch.int64), 'ws': e(int(need.value), torch.uint8)}
        B_.call('cdr_bpr_shard_plan', B_.ctx(self.device), B_.stream(), B_.i64(recv3), Bl, int(user_rows), int(item_local_rows), int(world),
                B_.i64(buf['u_loc']), B_.raw(buf['keys']), B_.raw(buf['perm']), B_.raw(buf['flags']), B_.raw(buf['heads']), B_.raw(buf['uidx']),
                B_.i64(buf['uniq_local']), B_.i64(buf['umap']), B_.i64(buf['counts']), B_.i64(buf['n_uniq']), B_.raw(buf['ws']), buf['ws'].numel())
        return {'Bl': Bl, 'buf': buf, 'u_loc': buf['u_loc'][:Bl], 'uniq_local': buf['uniq_local'], 'umap': buf['umap'][:2 * Bl],
                'counts': buf['counts'][:world]}

    def gather_rows_norms(self, table, local_ids):
        """The owner's side: (rows [n, D], their squared norms [n])."""
        B_ = self.B_
        n, D = local_ids.numel(), table.shape[1]
        rows = torch.empty(n, D, device=table.device, dtype=torch.float32)
        nrm2 = torch.empty(n, device=table.device, dtype=torch.float32)
        if n:
            B_.call('cdr_gather_rows_norms', B_.stream(), B_.f32(table), D, B_.i64(local_ids), n, B_.f32(rows), B_.f32(nrm2))
        return rows, nrm2

    def norm_sums(self, utab, plan, nrm2, sums3):
        """sums3 <- {0, sum ||U[u]||^2, sum nrm2[ip]} of this rank's routed triples."""
        B_ = self.B_
        B_.call('cdr_shard_norm_sums', B_.ctx(self.device), B_.stream(), B_.f32(utab), utab.shape[1], B_.i64(plan['u_loc']), B_.f32(nrm2),
                B_.i64(plan['umap']), plan['Bl'], B_.f32(sums3))

    def shard_step(self, utab, ustate, irows, plan, n_uniq, B_mean, gamma, reg_weight, opt, hp, step, out):
        """The requester's half in one call; returns GS [n_uniq, D]: ONE finished gradient row per distinct item (EmbLoss term inside).
        ``out[4:6]`` = the coefficients (in); ``out[6]`` = this rank's loss sum (out)."""
        B_ = self.B_
        buf, Bl, D = plan['buf'], plan['Bl'], utab.shape[1]
        GS = torch.empty(n_uniq, D, device=utab.device, dtype=torch.float32)
        gkey = ('gugp', torch.cuda.current_stream().cuda_stream, utab.data_ptr())
        gg = self._ws.get(gkey)
        if gg is None or gg[0].shape[0] < Bl:
            cap = max(int(Bl * 1.25), 1024)
            gg = self._ws[gkey] = (torch.empty(cap, D, device=utab.device, dtype=torch.float32), torch.empty(cap, D, device=utab.device, dtype=torch.float32))
        m, v = ustate if ustate is not None else (None, None)
        B_.call('cdr_bpr_shard_step', B_.ctx(self.device), B_.stream(), opt, B_.f32(utab), B_.f32(m), B_.f32(v), B_.f32(irows), D,
                B_.i64(plan['u_loc']), B_.i64(plan['umap']), Bl, int(B_mean), float(gamma), float(reg_weight), float(hp['lr']), float(hp['b1']),
                float(hp['b2']), float(hp['eps']), float(hp['wd']), int(step), B_.f32(out), B_.f32(gg[0]), B_.f32(gg[1]), B_.f32(GS),
                B_.raw(buf['keys']), B_.raw(buf['perm']), B_.raw(buf['flags']), B_.raw(buf['heads']), B_.raw(buf['uidx']))
        return GS

    def owner_apply(self, table, state, ids, runs, grads, opt, hp, step):
        """``ids``: ``runs`` ascending duplicate-free runs of local rows (one per requesting rank); ``grads`` one summed gradient row per id."""
        B_ = self.B_
        n = ids.numel()
        if n == 0:
            return
        dev = table.device
        need = ctypes.c_size_t(0)
        if runs > 1:
            B_._check(B_.load().cdr_sort_workspace_bytes(n, table.shape[0], ctypes.byref(need)), 'cdr_sort_workspace_bytes')
        ws = self._workspace(('sort', torch.cuda.current_stream().cuda_stream), max(need.value, 256), dev)
        kp = self._workspace(('owner_kp', torch.cuda.current_stream().cuda_stream), 8 * n, dev)
        keys, perm = kp[:4 * n], kp[4 * n:8 * n]
        m, v = (state if state is not None else (None, None))
        B_.call('cdr_shard_owner_apply', B_.ctx(self.device), B_.stream(), opt, B_.f32(table), B_.f32(m), B_.f32(v), table.shape[0], table.shape[1],
                B_.i64(ids), n, int(runs), B_.f32(grads), float(hp['lr']), float(hp['b1']), float(hp['b2']), float(hp['eps']), float(hp['wd']),
                int(step), B_.raw(keys), B_.raw(perm), B_.raw(ws), ws.numel())


SELF_VIA_COLLECTIVE = bool(int(__import__('os').environ.get('CDR_A2A_SELF_VIA_RCCL', '0')))   # 1: the round-1..4 behaviour (A/B runs, RCCL bring-up on one GPU)


def _a2a(inp, in_splits, out_splits, group, trailing=()):
    """all-to-all(v) of rows.  With ONE rank nothing has to move: the buffer is handed on as it is (measured on one MI355X with
    --force-shard: 5.3 of the 10.3 ms of a step were RCCL's eight self-copies inside all_to_all_single, which say nothing about the
    algorithm's local cost -- with more ranks those bytes travel over xGMI).  CDR_A2A_SELF_VIA_RCCL=1 keeps the collective in the
    one-rank case (RCCL bring-up on a single GPU, A/B runs)."""
    if dist.get_world_size(group) == 1 and not SELF_VIA_COLLECTIVE:
        assert in_splits[0] == out_splits[0] == inp.shape[0]
        return inp
    out = torch.empty((sum(out_splits),) + tuple(trailing), device=inp.device, dtype=inp.dtype)
    dist.all_to_all_single(out, inp, output_split_sizes=out_splits, input_split_sizes=in_splits, group=group)
    return out


def _allreduce(t, group):
    """In-place sum over the ranks; with ONE rank there is nothing to add (and no cross-stream hand-over to RCCL's stream to pay for)."""
    if dist.get_world_size(group) > 1 or SELF_VIA_COLLECTIVE:
        dist.all_reduce(t, group=group)
    return t


class CabiComm:
    """The row shard's data-path exchanges through the C ABI's own communicator (cdr_comm_init + cdr_a2a_ids / cdr_a2a_rows /
    cdr_allreduce_sum_f32, include/cdr_hip.h family (10)) instead of torch.distributed: what a host that is not Python calls.
    The 128-byte id travels over the existing process group (the "host's own means"); the bucket COUNTS stay on that group too
    (control plane, a few integers).  One communicator per (domain, stream), as one process group per domain.  With one rank the
    calls are still issued (self send / receive inside one RCCL group): that is what a one-GPU box can test."""

    def __init__(self, group=None, device=None):
        import ctypes
        from . import binding as B_
        self.B_, self.group = B_, group
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        idb = (ctypes.c_ubyte * 128)()
        if self.rank == 0:
            B_.call('cdr_comm_unique_id', ctypes.cast(idb, ctypes.c_void_p))
        box = [bytes(idb)]
        if self.world > 1:
            dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
        idb = (ctypes.c_ubyte * 128).from_buffer_copy(box[0])
        self.comm = ctypes.c_void_p()
        if device is not None:
            torch.cuda.set_device(device)
        B_.call('cdr_comm_init', ctypes.cast(ctypes.pointer(self.comm), ctypes.c_void_p), self.rank, self.world, ctypes.cast(idb, ctypes.c_void_p))
        self.calls = {'cdr_a2a_ids': 0, 'cdr_a2a_rows': 0, 'cdr_allreduce_sum_f32': 0}

    def info(self):
        """(rank, world) as the communicator itself reports them (cdr_comm_info)."""
        import ctypes
        r, w = ctypes.c_int(-1), ctypes.c_int(-1)
        self.B_.call('cdr_comm_info', self.comm, ctypes.cast(ctypes.pointer(r), ctypes.c_void_p), ctypes.cast(ctypes.pointer(w), ctypes.c_void_p))
        return r.value, w.value

    def a2a(self, inp, in_splits, out_splits, trailing=()):
        import ctypes
        B_ = self.B_
        unit = 1
        for t in trailing:
            unit *= int(t)
        inp = inp.contiguous()
        out = torch.empty((sum(out_splits),) + tuple(trailing), device=inp.device, dtype=inp.dtype)
        if inp.dtype == torch.int64:
            sc = (ctypes.c_int64 * self.world)(*[int(c) * unit for c in in_splits])
            rc = (ctypes.c_int64 * self.world)(*[int(c) * unit for c in out_splits])
            B_.call('cdr_a2a_ids', self.comm, B_.stream(), B_.i64(inp), ctypes.cast(sc, ctypes.c_void_p), B_.i64(out), ctypes.cast(rc, ctypes.c_void_p))
            self.calls['cdr_a2a_ids'] += 1
        elif inp.dtype == torch.float32:
            sc = (ctypes.c_int64 * self.world)(*[int(c) for c in in_splits])
            rc = (ctypes.c_int64 * self.world)(*[int(c) for c in out_splits])
            B_.call('cdr_a2a_rows', self.comm, B_.stream(), B_.f32(inp), ctypes.cast(sc, ctypes.c_void_p), B_.f32(out), ctypes.cast(rc, ctypes.c_void_p), unit)
            self.calls['cdr_a2a_rows'] += 1
        else:
            raise TypeError('CabiComm.a2a: int64 ids or fp32 rows, got %s' % inp.dtype)
        return out

    def allreduce(self, t):
        assert t.dtype == torch.float32 and t.is_contiguous()
        self.B_.call('cdr_allreduce_sum_f32', self.comm, self.B_.stream(), self.B_.f32(t), t.numel())
        self.calls['cdr_allreduce_sum_f32'] += 1
        return t

    def close(self):
        if self.comm:
            self.B_.call('cdr_comm_destroy', self.comm)
            self.comm = None


def _gather_counts(counts, extra, group, world):
    """All ranks' bucket counts (+ one extra int per rank) -> list of tensors (no host sync yet)."""
    payload = torch.cat([counts, torch.tensor([extra], device=counts.device, dtype=torch.int64)])
    if world == 1 and not SELF_VIA_COLLECTIVE:
        return [payload]
    gathered = [torch.empty_like(payload) for _ in range(world)]
    dist.all_gather(gathered, payload, group=group)
    return gathered


class ShardedBPRStep:
    """Multi-GPU counterpart of fused.FusedBPRStep.  ``user_shard`` / ``item_shard`` are this rank's rows
    (row r % G == rank, local index r // G)."""

    def __init__(self, user_shard, item_shard, n_users_total, n_items_total, max_batch, opt='adam', lr=1e-3,
                 betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, gamma=1e-10, reg_weight=0.0, group=None, ops=None,
                 stream=None, user_state=None, item_state=None, dedup=True, fuse_singles=True, comm=None, direct=True, index_group=None):
        from .fused import RowwiseState
        self.group = group
        # the id-only stages' own process group (= communicator): prefetched on a side stream they must not queue behind the main stage's
        # row exchanges on one RCCL stream (None: the step's group; fine on one rank and under gloo)
        self.index_group = index_group if index_group is not None else group
        self.comm = comm                          # optional CabiComm: rows / ids / sums travel through the C ABI's communicator
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
        # per-table optimizer state (moments + update count); may be shared with the OVERLAP phase's FusedMapStep
        self.ustate = user_state if user_state is not None else RowwiseState(user_shard, self.opt)
        self.istate = item_state if item_state is not None else RowwiseState(item_shard, self.opt)
        self.max_batch = int(max_batch)
        self.out = torch.zeros(12, device=dev, dtype=torch.float32)
        self.stream = stream                      # optional torch.cuda.Stream for pipelined execution
        self.dedup = bool(dedup)                  # request each distinct item row once, return one summed gradient row per item
        # (with dedup) the requester's half on the one-GPU step's kernels: user rows occurring once updated by the forward pass itself
        self.fuse_singles = bool(fuse_singles) and self.D % 4 == 0 and self.D <= 256
        self.n_items_total = int(n_items_total)
        # round 6 (the default): one sort per rank and step, gradient rows of items that occur once written straight into their send slot,
        # the owners' squared row norms beside the rows, one sorted run applied without a sort
        self.direct = bool(direct) and self.dedup and self.fuse_singles and hasattr(self.ops, 'plan')

    def loss_value(self):
        return self.out[0]

    def profile(self, on=True):
        """Count the bytes this rank sends to OTHER ranks and bracket every all-to-all with HIP events on the step's stream
        (bench.py reports both for N > 1: what the links carried and how long the step waited on them)."""
        self._prof = {'bytes': 0, 'events': []} if on else None

    def exchange_stats(self):
        """-> (bytes sent to other ranks, milliseconds inside all-to-alls) since ``profile()``; synchronises."""
        if not self.__dict__.get('_prof'):
            return 0, 0.0
        torch.cuda.synchronize()
        return self._prof['bytes'], sum(a.elapsed_time(b) for a, b in self._prof['events'])

    def _a2a(self, inp, in_splits, out_splits, group, trailing=()):
        if self.comm is not None and inp.is_cuda:
            return self.comm.a2a(inp, in_splits, out_splits, trailing)
        return _a2a(inp, in_splits, out_splits, group, trailing)

    def _sum(self, t, group):
        if self.comm is not None and t.is_cuda:
            return self.comm.allreduce(t)
        return _allreduce(t, group)

    def _x(self, inp, in_splits, out_splits, group, trailing=()):
        prof = self.__dict__.get('_prof')
        if not prof or not inp.is_cuda:
            return self._a2a(inp, in_splits, out_splits, group, trailing)
        row = inp.element_size() * (inp.numel() // max(inp.shape[0], 1))
        prof['bytes'] += row * (sum(in_splits) - in_splits[self.rank])
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        out = self._a2a(inp, in_splits, out_splits, group, trailing)
        b.record()
        prof['events'].append((a, b))
        return out

    @staticmethod
    def _moments(st):
        return (st.exp_avg, st.exp_avg_sq) if st.exp_avg is not None else None

    def _on_stream(self):
        return torch.cuda.stream(self.stream) if self.stream is not None else contextlib.nullcontext()

    def step(self, uid, pid, nid, next_batch=None):
        for _ in self.step_gen(uid, pid, nid, next_batch=next_batch):
            pass
        return self.out

    def step_gen(self, uid, pid, nid, next_batch=None):
        """Generator form of one step: yields at the two points where the host must wait for bucket counts.
        ``next_batch`` (direct form): the following step's (uid, pid, nid), if known -- its id-only stages (routing, the triples' and the
        request lists' all-to-alls, the sort, both host waits) are run on a side stream right behind this step's kernels, which then contain
        no host wait at all; the tensors must stay unmodified until that step is called with them."""
        G, grp, ops = self.world, self.group, self.ops
        B = uid.numel()
        self.ustate.advance()                      # per table, on every rank (also one whose buckets came up empty)
        self.istate.advance()
        if self.direct:
            if uid.is_cuda and next_batch is not None:
                with self._on_stream():
                    self._step_start = torch.cuda.Event()
                    self._step_start.record(torch.cuda.current_stream(uid.device))
            ix = self._take_prefetched(uid, pid, nid)
            if ix is None:
                ix = yield from self._index_gen(uid, pid, nid, side=False)
            self._main(ix)
            if next_batch is not None:
                self._prefetched = yield from self._index_gen(*next_batch, side=True)
            return

        # ---- 0. triples travel to the owner of their user row ---------------------------------------------------
        with self._on_stream():
            perm0, counts0 = ops.route(uid, None, G)
            send3 = torch.stack((ops.permute(uid, None, perm0, G), ops.permute(pid, None, perm0, 1),
                                 ops.permute(nid, None, perm0, 1)), dim=1).contiguous()
            gathered = _gather_counts(counts0, B, grp, G)
        yield
        with self._on_stream():
            allc = torch.stack(gathered).tolist()                           # host sync #1
            t_send = [int(c) for c in allc[self.rank][:G]]
            t_recv = [int(allc[r][self.rank]) for r in range(G)]
            B_global = sum(int(allc[r][G]) for r in range(G))
            recv3 = self._x(send3, t_send, t_recv, grp, (3,))
            Bl = recv3.shape[0]                                              # triples whose user row is mine
            u_loc = recv3[:, 0].contiguous()
            p2, n2 = recv3[:, 1].contiguous(), recv3[:, 2].contiguous()

        if self.dedup:                              # (outside the stream context: the generator yields in there)
            yield from self._items_dedup(uid.device, Bl, B_global, u_loc, p2, n2)
            return
        with self._on_stream():
            # ---- 1. item rows: ids to their owners, rows back ---------------------------------------------------
            if Bl:
                perm1, counts1 = ops.route(p2, n2, G)
                i_local_sorted = ops.permute(p2, n2, perm1, G, flag_below=Bl)     # bit 62 tags the positive occurrences
                pos_i = ops.inverse_perm(perm1)
            else:
                perm1 = torch.empty(0, device=uid.device, dtype=torch.int32)
                counts1 = torch.zeros(G, device=uid.device, dtype=torch.int64)
                i_local_sorted = torch.empty(0, device=uid.device, dtype=torch.int64)
                pos_i = torch.empty(0, device=uid.device, dtype=torch.int64)
            gathered = _gather_counts(counts1, 0, grp, G)
        yield
        with self._on_stream():
            allc = torch.stack(gathered).tolist()                           # host sync #2
            i_send = [int(c) for c in allc[self.rank][:G]]
            i_recv = [int(allc[r][self.rank]) for r in range(G)]
            i_req = self._x(i_local_sorted, i_send, i_recv, grp)                # local item rows other ranks want (tagged)
            i_req_rows = i_req & TAG_MASK
            irows = self._x(ops.gather_rows(self.I, i_req_rows), i_recv, i_send, grp, (self.D,))

            # ---- 2. fused forward + compact gradients; global loss reduction ----------------------------------
            GU = torch.empty(max(Bl, 1), self.D, device=uid.device, dtype=torch.float32)
            gi = torch.empty(2 * Bl, self.D, device=uid.device, dtype=torch.float32)     # send buffer, owner order
            if Bl:
                ops.fwd_grad(self.U, irows, u_loc, pos_i[:Bl].contiguous(), pos_i[Bl:].contiguous(), B_global, self.gamma,
                             self.reg_weight, self.out, GU, gi)
                sums = self.out[6:9].clone()
            else:
                sums = torch.zeros(3, device=uid.device, dtype=torch.float32)
            self._sum(sums, grp)
            ops.finish_sums(sums, B_global, self.reg_weight, self.out)

            # ---- 3. user rows are local; item gradients go home ---------------------------------------------------
            if Bl:
                ops.sort_apply(self.U, self._moments(self.ustate), u_loc, GU[:Bl], self.opt, self.hp, self.ustate.step,
                               reg_limit=Bl, reg_coef=self.out[4:5])
            gi_recv = self._x(gi, i_send, i_recv, grp, (self.D,))
            # the owner adds the EmbLoss term itself (it holds the pre-step row; the tag tells it which occurrences count)
            ops.sort_apply(self.I, self._moments(self.istate), i_req, gi_recv, self.opt, self.hp, self.istate.step,
                           reg_coef=self.out[5:6], tagged=True)


    # ---- the direct form (round 6): an id-only index stage (prefetchable) + a main stage without host waits -------------------------
    def _take_prefetched(self, uid, pid, nid):
        ix = self.__dict__.pop('_prefetched', None)
        if ix is not None and ix['key'] == (uid.data_ptr(), pid.data_ptr(), nid.data_ptr(), uid.numel()):
            return ix
        return None

    def _side_stream(self, dev):
        if self.__dict__.get('_istream') is None:
            self._istream = torch.cuda.Stream(device=dev, priority=int(__import__('os').environ.get('CDR_SIDE_PRIO', '0')))
            self._slot_free = [None, None]         # events: the main stage that last read plan slot k has been enqueued up to here
        return self._istream

    def _index_gen(self, uid, pid, nid, side):
        """Everything of a step that needs only its ids: triples to the owners of their user rows, ONE sort per rank (user rows | item keys
        grouped by owner) -> flags, duplicate heads, the request list; the request lists to the item owners.  Yields at the two host waits
        (bucket counts).  ``side``: on the step's side stream, into the plan slot the running main stage does not read."""
        G, grp, ops = self.world, self.index_group, self.ops
        dev, B = uid.device, uid.numel()
        cuda = uid.is_cuda
        self._slot = slot = 1 - self.__dict__.get('_slot', 1)
        # (a C-ABI communicator is one RCCL communicator: its calls must not run on two streams at once -> no side stream with it)
        side = side and cuda and self.comm is None
        if side:
            main = self.stream if self.stream is not None else torch.cuda.current_stream(dev)
            ist = self._side_stream(dev)
            # behind whatever the caller enqueued before this step (the producer of the next batch), and behind the main stage that last
            # read this plan slot -- NOT behind the main stage that has just been enqueued: that is the overlap
            ist.wait_event(self._step_start)
            ist.wait_stream(main) if self._slot_free[slot] is None else ist.wait_event(self._slot_free[slot])
            on = lambda: torch.cuda.stream(ist)  # noqa: E731
        else:
            on = self._on_stream
        with on():
            send3, counts0 = ops.route_triples(uid, pid, nid, G)
            gathered = _gather_counts(counts0, B, grp, G)
        yield
        with on():
            allc = torch.stack(gathered).tolist()                           # host wait #1
            t_send = [int(c) for c in allc[self.rank][:G]]
            t_recv = [int(allc[r][self.rank]) for r in range(G)]
            B_global = sum(int(allc[r][G]) for r in range(G))
            recv3 = self._x(send3, t_send, t_recv, grp, (3,))
            Bl = recv3.shape[0]                                              # triples whose user row is mine
            if Bl:
                plan = ops.plan(recv3, G, self.U.shape[0], shard_rows(self.n_items_total, G, 0), self.D, slot=slot)
                counts1 = plan['counts']
            else:
                plan, counts1 = None, torch.zeros(G, device=dev, dtype=torch.int64)
            gathered = _gather_counts(counts1, 0, grp, G)
        yield
        with on():
            allc = torch.stack(gathered).tolist()                           # host wait #2
            i_send = [int(c) for c in allc[self.rank][:G]]
            i_recv = [int(allc[r][self.rank]) for r in range(G)]
            n_uniq = sum(i_send)
            uniq = plan['uniq_local'][:n_uniq] if Bl else torch.empty(0, device=dev, dtype=torch.int64)
            i_req = self._x(uniq, i_send, i_recv, grp)                          # my item rows other ranks want, each once per rank
            done = None
            if side:
                done = torch.cuda.Event()
                done.record(ist)
                for t in (recv3, i_req):                                        # allocated on the side stream, read by the main stage
                    t.record_stream(main)
        return {'key': (uid.data_ptr(), pid.data_ptr(), nid.data_ptr(), B), 'hold': (uid, pid, nid, recv3), 'Bl': Bl, 'B_global': B_global,
                'plan': plan, 'i_send': i_send, 'i_recv': i_recv, 'n_uniq': n_uniq, 'i_req': i_req, 'done': done, 'slot': slot}

    def _main(self, ix):
        """Rows out, rows back, forward-and-update, gradient rows home, owner apply -- no host wait.  The owners send the rows with their
        squared norms (the EmbLoss norm needs no second pass over the received rows); the forward-and-update pass writes the finished gradient
        row of every item that occurs once straight into its send slot and only duplicate items go through a segmented sum; an owner that was
        asked by one rank applies the (already ascending, duplicate-free) list as it stands."""
        G, grp, ops = self.world, self.group, self.ops
        Bl, B_global, plan, i_send, i_recv, n_uniq, i_req = (ix[k] for k in ('Bl', 'B_global', 'plan', 'i_send', 'i_recv', 'n_uniq', 'i_req'))
        dev = i_req.device
        with self._on_stream():
            if ix['done'] is not None:
                torch.cuda.current_stream(dev).wait_event(ix['done'])
            rows, nrm2 = ops.gather_rows_norms(self.I, i_req)
            irows = self._x(rows, i_recv, i_send, grp, (self.D,))
            inrm = self._x(nrm2, i_recv, i_send, grp)
            sums = self.out[6:9]                                                # {loss sum, sum u^2, sum p^2}: all-reduced in place
            if Bl:
                ops.norm_sums(self.U, plan, inrm, sums)
            else:
                sums.zero_()
            self._sum(sums, grp)
            ops.finish_sums(sums, B_global, self.reg_weight, self.out)          # out[4:6] = the coefficients every rank uses
            if Bl:
                gi = ops.shard_step(self.U, self._moments(self.ustate), irows, plan, n_uniq, B_global, self.gamma, self.reg_weight, self.opt,
                                    self.hp, self.ustate.step, self.out)        # out[6] = this rank's loss sum
            else:
                gi = torch.empty(0, self.D, device=dev, dtype=torch.float32)
                self.out[6:7].zero_()
            self._sum(self.out[6:7], grp)
            ops.finish_sums(sums, B_global, self.reg_weight, self.out)
            gi_recv = self._x(gi, i_send, i_recv, grp, (self.D,))
            ops.owner_apply(self.I, self._moments(self.istate), i_req, G, gi_recv, self.opt, self.hp, self.istate.step)
            if self.__dict__.get('_istream') is not None:
                ev = torch.cuda.Event()
                ev.record(torch.cuda.current_stream(dev))
                self._slot_free[ix['slot']] = ev

    def _items_dedup(self, dev, Bl, B_global, u_loc, p2, n2):
        """Steps 1-3 with id de-duplication: every distinct item row crosses xGMI once per step in each direction (the rows
        out, ONE gradient row -- summed over the item's occurrences here, EmbLoss term included -- back)."""
        G, grp, ops = self.world, self.group, self.ops
        with self._on_stream():
            if Bl:
                plan = ops.dedup(p2, n2, G, shard_rows(self.n_items_total, G, 0))
                counts1 = plan['counts']
            else:
                plan, counts1 = None, torch.zeros(G, device=dev, dtype=torch.int64)
            gathered = _gather_counts(counts1, 0, grp, G)
        yield
        with self._on_stream():
            allc = torch.stack(gathered).tolist()                           # host sync #2
            i_send = [int(c) for c in allc[self.rank][:G]]
            i_recv = [int(allc[r][self.rank]) for r in range(G)]
            n_uniq = sum(i_send)
            uniq = plan['uniq_local'][:n_uniq] if Bl else torch.empty(0, device=dev, dtype=torch.int64)
            i_req = self._x(uniq, i_send, i_recv, grp)                          # my item rows other ranks want, each once per rank
            irows = self._x(ops.gather_rows(self.I, i_req), i_recv, i_send, grp, (self.D,))
            if self.fuse_singles and hasattr(ops, 'local_step'):
                # round 5: the one-GPU step's forward-and-update pass on the requester's side.  The EmbLoss coefficients need the norms of
                # the GLOBAL batch before the first row moves: a norms pass over the rows held here + a 12-byte all-reduce come first.
                if Bl:
                    umap = plan['umap']
                    ip, in_ = umap[:Bl].contiguous(), umap[Bl:].contiguous()
                    norms = ops.batch_norms(self.U, irows, u_loc, ip)
                else:
                    norms = torch.zeros(3, device=dev, dtype=torch.float32)
                self._sum(norms, grp)
                ops.finish_sums(norms, B_global, self.reg_weight, self.out)          # out[4:6] = the coefficients every rank uses
                if Bl:
                    GP = ops.local_step(self.U, self._moments(self.ustate), irows, u_loc, ip, in_, B_global, self.gamma, self.reg_weight,
                                        self.opt, self.hp, self.ustate.step, self.out)
                    loss = self.out[6:7].clone()
                else:
                    loss = torch.zeros(1, device=dev, dtype=torch.float32)
                self._sum(loss, grp)
                norms[0:1] = loss                                                    # {global loss sum, global sum u^2, global sum p^2}
                ops.finish_sums(norms, B_global, self.reg_weight, self.out)
                gi = ops.segsum(plan, GP[:Bl], Bl, irows, self.out[5:6], n_uniq) if Bl else torch.empty(0, self.D, device=dev, dtype=torch.float32)
            else:
                GU = torch.empty(max(Bl, 1), self.D, device=dev, dtype=torch.float32)
                GP = torch.empty(max(Bl, 1), self.D, device=dev, dtype=torch.float32)
                if Bl:
                    umap = plan['umap']
                    ops.fwd_grad(self.U, irows, u_loc, umap[:Bl].contiguous(), umap[Bl:].contiguous(), B_global, self.gamma,
                                 self.reg_weight, self.out, GU, GP, scatter=False)
                    sums = self.out[6:9].clone()
                else:
                    sums = torch.zeros(3, device=dev, dtype=torch.float32)
                self._sum(sums, grp)
                ops.finish_sums(sums, B_global, self.reg_weight, self.out)
                if Bl:
                    ops.sort_apply(self.U, self._moments(self.ustate), u_loc, GU[:Bl], self.opt, self.hp, self.ustate.step,
                                   reg_limit=Bl, reg_coef=self.out[4:5])
                    gi = ops.segsum(plan, GP[:Bl], Bl, irows, self.out[5:6], n_uniq)
                else:
                    gi = torch.empty(0, self.D, device=dev, dtype=torch.float32)
            gi_recv = self._x(gi, i_send, i_recv, grp, (self.D,))
            # the EmbLoss term is already inside the rows: the owner just sums what it received per row and applies
            ops.sort_apply(self.I, self._moments(self.istate), i_req, gi_recv, self.opt, self.hp, self.istate.step)


def run_pipelined(generators):
    """Round-robin several ``step_gen`` generators until all finish: while one waits for its bucket counts the others'
    kernels and collectives are already enqueued on their own streams."""
    alive = list(generators)
    while alive:
        for g in list(alive):
            try:
                next(g)
            except StopIteration:
                alive.remove(g)


class ShardedFullSort:
    """``full_sort_predict`` (emcdr.py:208-233, TARGET / OVERLAP phase) over row-sharded tables: every rank scores the
    eval users against ITS rows of the target item table (the 1-GPU scoring kernels on N/G rows), the partial score
    matrices are all-gathered, and one streaming kernel puts them into the reference's ``[U, N]`` item-id order.

    ``n_scored`` = the number of leading global item rows that are scored (``target_num_items`` = OI + TOI: the target
    phase scores a prefix of the union-sized table).  The eval users' rows are replicated first (``user_rows``): U*D*4
    bytes, nothing next to the U*N*4*(G-1)/G bytes of the score all-gather, which is what bounds this path on xGMI --
    callers that only need top-k should rank per shard and exchange k candidates instead."""

    def __init__(self, item_shard, n_scored, group=None):
        from . import binding as B_
        self.B_ = B_
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.I = item_shard
        self.N = int(n_scored)
        self.Nl = (self.N + self.world - 1) // self.world            # rows per shard in the gathered layout (last one padded)
        self._pad = torch.zeros(1, item_shard.shape[1], device=item_shard.device, dtype=torch.float32)

    def user_rows(self, table_shard, ids):
        """Rows ``ids`` (replicated list of GLOBAL ids) of a sharded table, replicated on every rank."""
        B_ = self.B_
        ids = ids.reshape(-1).contiguous()
        out = torch.empty(ids.numel(), table_shard.shape[1], device=table_shard.device, dtype=torch.float32)
        B_.call('cdr_gather_owned_rows', B_.stream(), B_.f32(table_shard), table_shard.shape[1], B_.i64(ids), ids.numel(),
                self.world, self.rank, B_.f32(out))
        dist.all_reduce(out, group=self.group)
        return out

    def local_scores(self, user_e):
        """[U, Nl] scores of this rank's item rows r = rank, rank + G, ... < n_scored (column Nl-1 may be padding)."""
        from . import functional as F_
        mine = shard_rows(self.N, self.world, self.rank)
        if mine == self.Nl:
            return F_.fullsort_scores(user_e, self.I[:mine])
        return F_.fullsort_scores(user_e, self.I[:mine], self._pad)                  # mine == Nl - 1: one padded column

    def scores(self, user_e):
        """user_e [U, D] (replicated) -> [U, n_scored] on every rank, identical to the single-device result."""
        B_ = self.B_
        U = user_e.shape[0]
        part = self.local_scores(user_e)
        gathered = torch.empty(self.world * U, self.Nl, device=part.device, dtype=torch.float32)   # [G][U][Nl], dim-0 concat
        dist.all_gather_into_tensor(gathered, part, group=self.group)
        out = torch.empty(U, self.N, device=part.device, dtype=torch.float32)
        B_.call('cdr_interleave_shards', B_.stream(), B_.f32(gathered), self.world, U, self.Nl, self.N, B_.f32(out))
        return out

    def topk(self, user_e, k, hist_indptr=None, hist_cols=None, exclude_first_col=True):
        """(values, global columns) [U, k] of ``scores(user_e)`` after the evaluation mask, identical on every rank, WITHOUT
        the all-gather of the score matrix: each rank runs the fused mask + top-k kernel over its own item rows and only
        its k candidates per user (12 B each) are exchanged -- the form that scales.
        ``hist_indptr`` / ``hist_cols``: per-user ascending GLOBAL columns to skip (replicated)."""
        from . import functional as F_
        G, rank = self.world, self.rank
        U = user_e.shape[0]
        mine = shard_rows(self.N, G, rank)
        lp = lc = None
        if hist_indptr is not None:
            own = (hist_cols % G) == rank                                    # my share of every user's history, local columns
            lc = (hist_cols[own] // G).contiguous()
            owner_user = torch.repeat_interleave(torch.arange(U, device=hist_cols.device), hist_indptr[1:] - hist_indptr[:-1])
            lp = torch.zeros(U + 1, device=hist_cols.device, dtype=torch.int64)
            lp[1:] = torch.cumsum(torch.bincount(owner_user[own], minlength=U), 0)
            if lc.numel() == 0:
                lp = lc = None
        kk = min(k, mine)
        vals, lidx = F_.fullsort_topk(user_e, self.I[:mine], None, k=kk, hist_indptr=lp, hist_cols=lc,
                                      exclude_first_col=bool(exclude_first_col and rank == 0))
        if kk < k:
            vals = torch.cat([vals, vals.new_full((U, k - kk), float('-inf'))], 1)
            lidx = torch.cat([lidx, lidx.new_full((U, k - kk), -1)], 1)
        allv = torch.empty(G * U, k, device=vals.device, dtype=torch.float32)             # [G][U][k], dim-0 concat
        alli = torch.empty(G * U, k, device=vals.device, dtype=torch.int64)
        dist.all_gather_into_tensor(allv, vals.contiguous(), group=self.group)
        dist.all_gather_into_tensor(alli, lidx.contiguous(), group=self.group)
        B_ = self.B_
        out_v = torch.empty(U, k, device=vals.device, dtype=torch.float32)
        out_i = torch.empty(U, k, device=vals.device, dtype=torch.int64)
        B_.call('cdr_topk_merge_shards', B_.stream(), B_.f32(allv), B_.i64(alli), G, U, k, 1, B_.f32(out_v), B_.i64(out_i))
        return out_v, out_i


    def topk_ranges(self, user_e, k, ranges, hist_indptr=None, hist_cols=None, exclude_first_col=True):
        """``topk`` when the scored slab is the concatenation of up to two row RANGES of the sharded table -- the SOURCE phase
        of emcdr.py:208-214 scores ``cat(W_s[:OI], W_s[TI:])``.  ``ranges`` = [(lo, hi), ...] in concatenation order; output
        columns (and ``hist_cols``) count along the concatenation, as ``full_sort_predict``'s do.  Each rank scores its rows of
        every range (a contiguous local slice each), converts its candidates to output columns, and the lists are merged."""
        from . import functional as F_
        B_ = self.B_
        G, rank = self.world, self.rank
        U = user_e.shape[0]
        assert 1 <= len(ranges) <= 2
        # local slice of range j: global row r = l * G + rank in [lo, hi)  <=>  l in [ceil((lo - rank) / G), ceil((hi - rank) / G))
        loc = [((lo - rank + G - 1) // G if lo > rank else 0, (hi - rank + G - 1) // G if hi > rank else 0) for lo, hi in ranges]
        base = [0]
        for lo, hi in ranges:
            base.append(base[-1] + (hi - lo))                                   # output column of each range's first row
        lbase = [0]
        for a, b in loc:
            lbase.append(lbase[-1] + (b - a))                                   # local column of each range's first local row
        slabs = [self.I[a:b] for a, b in loc]
        lp = lc = None
        if hist_indptr is not None and hist_cols.numel():
            c = hist_cols
            j = torch.zeros_like(c) if len(ranges) == 1 else (c >= base[1]).long()
            lo_t = torch.tensor([r[0] for r in ranges], device=c.device)[j]
            r = lo_t + (c - torch.tensor(base[:-1], device=c.device)[j])        # global row of each history column
            own = (r % G) == rank
            lcol = (r // G) - torch.tensor([a for a, _ in loc], device=c.device)[j] + torch.tensor(lbase[:-1], device=c.device)[j]
            owner_user = torch.repeat_interleave(torch.arange(U, device=c.device), hist_indptr[1:] - hist_indptr[:-1])
            key = torch.sort(owner_user[own] * (lbase[-1] + 1) + lcol[own]).values      # ascending local columns per user
            lc = (key % (lbase[-1] + 1)).contiguous()
            lp = torch.zeros(U + 1, device=c.device, dtype=torch.int64)
            lp[1:] = torch.cumsum(torch.bincount(key // (lbase[-1] + 1), minlength=U), 0)
            if lc.numel() == 0:
                lp = lc = None
        n_local = lbase[-1]
        kk = min(k, n_local)
        first_mine = exclude_first_col and ranges[0][0] % G == rank and loc[0][1] > loc[0][0]
        if kk > 0:
            s0 = slabs[0] if slabs[0].shape[0] else None
            s1 = slabs[1] if len(slabs) > 1 and slabs[1].shape[0] else None
            if s0 is None:
                s0, s1 = s1, None
            vals, lidx = F_.fullsort_topk(user_e, s0, s1, k=kk, hist_indptr=lp, hist_cols=lc, exclude_first_col=bool(first_mine))
        else:
            vals = user_e.new_empty(U, 0)
            lidx = torch.empty(U, 0, device=user_e.device, dtype=torch.int64)
        if kk < k:
            vals = torch.cat([vals, vals.new_full((U, k - kk), float('-inf'))], 1)
            lidx = torch.cat([lidx, lidx.new_full((U, k - kk), -1)], 1)
        # local column -> output column (index plumbing): range j, local row l = loc[j][0] + (col - lbase[j]), row r = l*G + rank
        col = lidx.clamp(min=0)
        j = torch.zeros_like(col) if len(ranges) == 1 else (col >= lbase[1]).long()
        t = lambda xs: torch.tensor(xs, device=col.device)[j]
        r = (t([a for a, _ in loc]) + (col - t(lbase[:-1]))) * G + rank
        out_col = torch.where(lidx >= 0, t(base[:-1]) + (r - t([lo for lo, _ in ranges])), lidx)
        allv = torch.empty(G * U, k, device=vals.device, dtype=torch.float32)
        alli = torch.empty(G * U, k, device=vals.device, dtype=torch.int64)
        dist.all_gather_into_tensor(allv, vals.contiguous(), group=self.group)
        dist.all_gather_into_tensor(alli, out_col.contiguous(), group=self.group)
        out_v = torch.empty(U, k, device=vals.device, dtype=torch.float32)
        out_i = torch.empty(U, k, device=vals.device, dtype=torch.int64)
        B_.call('cdr_topk_merge_shards', B_.stream(), B_.f32(allv), B_.i64(alli), G, U, k, 0, B_.f32(out_v), B_.i64(out_i))
        return out_v, out_i
