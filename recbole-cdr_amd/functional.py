"""torch.autograd bridges onto libcdrhip: each Function's forward/backward is one or two native kernel launches.

The reference's caller (recbole ``Trainer._train_epoch``) runs ``loss.backward()`` and a dense ``torch.optim.Adam``
over ``model.parameters()``, so in drop-in mode the backward kernels write the dense ``[rows, D]`` gradients that a
``sparse=False`` ``nn.Embedding`` would receive.  The fused row-wise training step that never
materialises table-sized gradients lives in ``fused.py``.
"""
import ctypes
import os

import torch
from torch.autograd import Function

from . import binding as B_


def _zeros_like2(a, b):
    """Two zero tensors shaped like a and b out of ONE buffer: one fill launch instead of two (each launch is a tenth of the C2 step)."""
    if b is None:
        return torch.zeros_like(a), None
    flat = torch.zeros(a.numel() + b.numel(), device=a.device, dtype=torch.float32)
    return flat[:a.numel()].view(a.shape), flat[a.numel():].view(b.shape)


def _prezero_for_backward(ctx, *tables, lists=None):
    """Called in a loss's forward, BEFORE its forward launch: when a backward will follow, allocate the dense gradient buffers of
    ``tables`` now (one flat buffer) and let the forward launch zero-fill them on the side (cdr_ctx_scrub_next) -- the separate fill
    launch was ~5 us of a 40 us step at the reference's default batch.  ``_grads_for`` hands them out in the backward."""
    ctx.pre = None
    tabs = [t for t in tables if t is not None]
    if (deterministic() and not ordered_backward()) or not tabs or not any(ctx.needs_input_grad) or not all(t.is_contiguous() for t in tabs):
        return
    # under set_deterministic the buffers are consumed by the ordered launch only; lists past its cap take the sorted route, which brings its
    # own buffers: do not allocate and scrub table-sized memory for nothing (``lists`` = the occurrence counts per gradient buffer)
    if deterministic() and lists is not None and not ordered_fits(tabs[0].shape[1], *lists):
        return
    offs, n = [], 0
    for t in tabs:
        offs.append(n)
        n += (t.numel() + 3) // 4 * 4                       # every view starts on a 16-byte boundary
    flat = torch.empty(n, device=tabs[0].device, dtype=torch.float32)
    B_.call('cdr_ctx_scrub_next', B_.ctx(flat.device), B_.raw(flat), 4 * n)
    ctx.pre = [flat[o:o + t.numel()].view(t.shape) for o, t in zip(offs, tabs)]


def _grads_for(ctx, a, b):
    """Zeroed gradient buffers shaped like a and b: the ones the forward launch cleared (first backward through this node), else a
    fresh fill."""
    pre = getattr(ctx, 'pre', None)
    if pre is not None and len(pre) == (1 if b is None else 2):
        ctx.pre = None
        return pre[0], (pre[1] if b is not None else None)
    return _zeros_like2(a, b)


# ---- run-to-run reproducible dense gradients (opt-in) -------------------------------------------------------------------------------
# The drop-in losses hand autograd DENSE [rows, D] gradients built with fp32 atomics, like torch's embedding backward: the order of the
# adds into a row that occurs several times in a batch is not fixed, so two runs can differ in the last bit.  With
# ``set_deterministic(True)`` (or CDR_DETERMINISTIC=1) the same kernels run on the batch's GATHERED rows with the occurrence index as
# the id -- every address receives exactly one add -- and the per-occurrence rows are then summed per distinct id in occurrence order
# (id sort + cdr_scatter_rows_sorted, as the CoNet node always does).  Covers EMCDR (BPR and MF), CMF, SSCDR and BiTGCF; a few launches more
# per step, so it is off by default.
_DETERMINISTIC = [False]


def set_deterministic(flag):
    _DETERMINISTIC[0] = bool(flag)


def deterministic():
    return _DETERMINISTIC[0] or os.environ.get('CDR_DETERMINISTIC', '0') == '1'


# ---- ordered dense backward: what ``set_deterministic(True)`` runs at the reference's batch sizes ----------------------------------
# cdr_ordered_bwd (csrc/cdr_ordered.hip): every list of occurrences that adds into one gradient buffer is walked by ONE launch that
# finds each row's occurrences by an all-pairs id test out of LDS and sums their terms in list order -- the occurrence-order sum of
# the sorted form above without the gathers, the per-occurrence rows, the id sort and the segmented scatter: one launch instead of
# 10-20.  It is quadratic in the list length, so it takes lists of up to ``ordered_max()`` entries (8,192: twice the item list of a
# 2,048-triple BPR batch, overall.yaml:19, and where it still beats the sorted form; env CDR_ORDERED_MAX, at most 16,384; 0 or
# ``set_ordered_backward(False)`` switches it off) and longer lists keep the sorted form.  Measured (tools/mb_ordered_bwd.py,
# profiles/r05_mb_ordered_bwd.txt): forward + backward of a 2,048-triple BPR batch 36 us against 55 sorted (25 with atomics), CMF's two
# domains 50 against 87 (33); 4,096 triples 50 against 67; 8,192 triples (lists of 16,384) 116 against 101 -- hence the cap.
_ORDERED = [True]
_ORDERED_MAX = [None]


def set_ordered_backward(flag, max_entries=None):
    _ORDERED[0] = bool(flag)
    _ORDERED_MAX[0] = None if max_entries is None else int(max_entries)


def ordered_max():
    if _ORDERED_MAX[0] is not None:
        return max(0, min(int(_ORDERED_MAX[0]), B_.ORD_MAX_TOTAL))
    return max(0, min(int(os.environ.get('CDR_ORDERED_MAX', '8192')), B_.ORD_MAX_TOTAL))


def ordered_backward():
    return _ORDERED[0] and ordered_max() > 0


def ordered_fits(D, *totals):
    """Under ``set_deterministic`` the ordered launch takes these lists: float4 rows of at most 256 floats, every list within the cap."""
    if not deterministic() or not ordered_backward() or D % 4 != 0 or D > 256:
        return False
    cap = ordered_max()
    return all(0 <= int(t) <= cap for t in totals) and any(int(t) > 0 for t in totals)


_ord_fits = ordered_fits


def _ptr(t, byte_off=0):
    return None if t is None else t.data_ptr() + byte_off


def _ord_seg(ids, coef=None, sign=1.0, go=None, go_scale=1.0, X=None, xid=None, Y=None, yid=None, x_stride=0,
             R=None, r_stride=0, norm=None, reg=0.0, B=1):
    """One segment of a list (cdr_ord_seg); pointer arguments are integers (``tensor.data_ptr()`` + byte offset) or None."""
    sg = B_.OrdSeg()
    sg.ids, sg.n = ids.data_ptr(), ids.numel()
    sg.coef, sg.sign, sg.go, sg.go_scale = coef, float(sign), go, float(go_scale)
    sg.X, sg.xid, sg.Y, sg.yid, sg.x_stride = X, xid, Y, yid, int(x_stride)
    sg.R, sg.r_stride, sg.norm, sg.reg_weight, sg.B = R, int(r_stride), norm, float(reg), int(B)
    return sg


def _ordered_bwd(D, lists, keep=(), accumulate=False):
    """lists: [(g pointer, g row stride, [segments])] -- one launch; ``keep`` holds the operands until it is enqueued.  accumulate: the
    sums are added to the rows' earlier content (buffers that already hold another node's gradient)."""
    lists = [l for l in lists if l[2]]
    if not lists:
        return
    arr = (B_.OrdList * len(lists))()
    for k, (g, gs, segs) in enumerate(lists):
        arr[k].g, arr[k].g_stride, arr[k].nseg, arr[k].accumulate = g, int(gs), len(segs), int(bool(accumulate))
        for q, sg in enumerate(segs):
            arr[k].seg[q] = sg
    B_.call('cdr_ordered_bwd', B_.stream(), int(D), arr, len(lists))
    del keep


_arange_cache = {}


def _arange(dev, n):
    key = str(dev)
    a = _arange_cache.get(key)
    if a is None or a.numel() < n:
        a = torch.arange(max(n, 4096), device=dev, dtype=torch.int64)
        _arange_cache[key] = a
    return a[:n]


def _gather_plain(weight, ids):
    out = torch.empty(ids.numel(), weight.shape[1], device=weight.device, dtype=torch.float32)
    if ids.numel():
        B_.call('cdr_gather_rows', B_.stream(), B_.f32(weight), weight.shape[1], B_.i64(ids), ids.numel(), B_.f32(out))
    return out


def _scatter_rows_deterministic(shape, ids, rows):
    """zeros(shape) with rows[o] added to row ids[o], the occurrences of a row summed in occurrence order (no float atomics)."""
    g = torch.zeros(shape, device=rows.device, dtype=torch.float32)
    if ids.numel():
        (k, p, n), = sort_id_lists([ids], shape[0])
        rows = rows.contiguous()
        B_.call('cdr_scatter_rows_sorted', B_.stream(), B_.f32(g), shape[1], B_.raw(k), B_.raw(p), n, B_.f32(rows), rows.shape[1])
    return g


def _det_point_rows(U, I, RU, RI, uid, iid, gcoef, out4_ptr, reg, go):
    """Per-occurrence gradient rows of one pointwise batch from cdr_point_bwd_dense on the gathered rows: (dU, dI, dRU, dRI), the last two
    None when the EmbLoss tables are the dot tables (their term is then inside dU / dI)."""
    dev, D, n = U.device, U.shape[1], uid.numel()
    ar = _arange(dev, n)
    sep_u = RU is not None and RU.data_ptr() != U.data_ptr()
    sep_i = RI is not None and RI.data_ptr() != I.data_ptr()
    Ur, Ir = _gather_plain(U, uid), _gather_plain(I, iid)
    RUr = _gather_plain(RU, uid) if sep_u else None
    RIr = _gather_plain(RI, iid) if sep_i else None
    flat = torch.zeros((2 + int(sep_u) + int(sep_i)) * n * D, device=dev, dtype=torch.float32)
    parts = [flat[k * n * D:(k + 1) * n * D].view(n, D) for k in range(2 + int(sep_u) + int(sep_i))]
    dU, dI = parts[0], parts[1]
    dRU = parts[2] if sep_u else None
    dRI = parts[2 + int(sep_u)] if sep_i else None
    B_.call('cdr_point_bwd_dense', B_.ctx(dev), B_.stream(), B_.f32(Ur), B_.f32(Ir), B_.f32(RUr), B_.f32(RIr), D, B_.i64(ar), B_.i64(ar), n,
            B_.f32(gcoef), out4_ptr, float(reg), B_.f32(go), B_.f32(dU), B_.f32(dI), B_.f32(dRU), B_.f32(dRI))
    return dU, dI, dRU, dRI


def _dev_check(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise B_.NativeLibraryError('recbole_cdr_amd runs on a ROCm device only: got a CPU tensor '
                                        '(move the model and the interaction to config["device"])')


def _ids(t):
    return t.reshape(-1).contiguous().to(torch.int64)


class BPRGatherLoss(Function):
    """emcdr.py:123-131 / :146-154 in one launch: gather x3 -> 2 dots -> BPRLoss + reg_weight * EmbLoss."""

    @staticmethod
    def forward(ctx, user_w, item_w, uid, pid, nid, gamma, reg_weight):
        _dev_check(user_w, item_w, uid, pid, nid)
        uid, pid, nid = _ids(uid), _ids(pid), _ids(nid)
        n, D = uid.numel(), user_w.shape[1]
        out4 = torch.empty(4, device=user_w.device, dtype=torch.float32)
        g = torch.empty(n, device=user_w.device, dtype=torch.float32)
        _prezero_for_backward(ctx, user_w, item_w, lists=(n, 2 * n))
        B_.call('cdr_bpr_fwd', B_.ctx(user_w.device), B_.stream(), B_.f32(user_w), B_.f32(item_w), D,
                B_.i64(uid), B_.i64(pid), B_.i64(nid), n, float(gamma), float(reg_weight), B_.f32(out4), B_.f32(g))
        ctx.save_for_backward(user_w, item_w, uid, pid, nid, g, out4)
        ctx.reg_weight = float(reg_weight)
        return out4[:1]

    @staticmethod
    def backward(ctx, grad_out):
        user_w, item_w, uid, pid, nid, g, out4 = ctx.saved_tensors
        go = grad_out.reshape(-1).contiguous().to(torch.float32)
        n, D = uid.numel(), user_w.shape[1]
        if _ord_fits(D, n, 2 * n) and user_w.is_contiguous() and item_w.is_contiguous():
            gU, gI = _grads_for(ctx, user_w, item_w)
            uw, iw, gp, o4, gop = user_w.data_ptr(), item_w.data_ptr(), g.data_ptr(), out4.data_ptr(), go.data_ptr()
            _ordered_bwd(D, [
                (gU.data_ptr(), D, [_ord_seg(uid, coef=gp, go=gop, X=iw, xid=pid.data_ptr(), Y=iw, yid=nid.data_ptr(), x_stride=D,
                                             R=uw, r_stride=D, norm=o4 + 8, reg=ctx.reg_weight, B=n)]),
                (gI.data_ptr(), D, [_ord_seg(pid, coef=gp, go=gop, X=uw, xid=uid.data_ptr(), x_stride=D,
                                             R=iw, r_stride=D, norm=o4 + 12, reg=ctx.reg_weight, B=n),
                                    _ord_seg(nid, coef=gp, sign=-1.0, go=gop, X=uw, xid=uid.data_ptr(), x_stride=D)])],
                keep=(go, gU, gI))
            return gU, gI, None, None, None, None, None
        if deterministic():
            dev, D, n = user_w.device, user_w.shape[1], uid.numel()
            ar = _arange(dev, 2 * n)
            items = torch.cat([pid, nid])
            Ur, Ir = _gather_plain(user_w, uid), _gather_plain(item_w, items)
            flat = torch.zeros(3 * n * D, device=dev, dtype=torch.float32)
            dU, dI = flat[:n * D].view(n, D), flat[n * D:].view(2 * n, D)
            B_.call('cdr_bpr_bwd_dense', B_.ctx(dev), B_.stream(), B_.f32(Ur), B_.f32(Ir), D, B_.i64(ar[:n]), B_.i64(ar[:n]), B_.i64(ar[n:]), n,
                    B_.f32(g), B_.f32(out4), ctx.reg_weight, B_.f32(go), B_.f32(dU), B_.f32(dI))
            return (_scatter_rows_deterministic(user_w.shape, uid, dU), _scatter_rows_deterministic(item_w.shape, items, dI),
                    None, None, None, None, None)
        gU, gI = _grads_for(ctx, user_w, item_w)
        B_.call('cdr_bpr_bwd_dense', B_.ctx(user_w.device), B_.stream(), B_.f32(user_w), B_.f32(item_w), user_w.shape[1],
                B_.i64(uid), B_.i64(pid), B_.i64(nid), uid.numel(), B_.f32(g), B_.f32(out4), ctx.reg_weight,
                B_.f32(go), B_.f32(gU), B_.f32(gI))
        return gU, gI, None, None, None, None, None


class PointGatherLoss(Function):
    """Pointwise gather -> dot -> (sigmoid) -> MSE/BCE + reg_weight * EmbLoss  (emcdr.py:111-122 ; cmf.py:90-98 ;
    bitgcf.py:221-247 with separate EmbLoss tables).  Returns (total[1], main[1], scores[B])."""

    @staticmethod
    def forward(ctx, kind, user_w, item_w, reg_user_w, reg_item_w, uid, iid, label, reg_weight):
        _dev_check(user_w, item_w, uid, iid, label)
        uid, iid = _ids(uid), _ids(iid)
        label = label.reshape(-1).contiguous().to(torch.float32)
        n, D = uid.numel(), user_w.shape[1]
        dev = user_w.device
        out4 = torch.empty(4, device=dev, dtype=torch.float32)
        g = torch.empty(n, device=dev, dtype=torch.float32)
        scores = torch.empty(n, device=dev, dtype=torch.float32)
        B_.call('cdr_point_fwd', B_.ctx(dev), B_.stream(), int(kind), B_.f32(user_w), B_.f32(item_w),
                B_.f32(reg_user_w), B_.f32(reg_item_w), D, B_.i64(uid), B_.i64(iid), B_.f32(label), n,
                float(reg_weight), B_.f32(out4), B_.f32(g), B_.f32(scores))
        ctx.save_for_backward(user_w, item_w, reg_user_w, reg_item_w, uid, iid, g, out4)
        ctx.reg_weight = float(reg_weight)
        ctx.mark_non_differentiable(scores)
        ctx.set_materialize_grads(False)      # no zero-filled gradient for the non-differentiable output (a launch per step)
        return out4[:1], scores

    @staticmethod
    def backward(ctx, grad_out, _gs):
        user_w, item_w, reg_user_w, reg_item_w, uid, iid, g, out4 = ctx.saved_tensors
        # the same tensor as user AND item operand (BiTGCF scores rows of one stacked [users ; items] table): one gradient buffer,
        # both scatters add into it, and autograd gets it once -- instead of two table-sized buffers plus the add that merges them
        shared = user_w.data_ptr() == item_w.data_ptr() and user_w.shape == item_w.shape and user_w.stride() == item_w.stride()
        n, D = uid.numel(), user_w.shape[1]
        if (_ord_fits(D, 2 * n if shared else n) and user_w.is_contiguous() and item_w.is_contiguous()
                and all(r is None or (r.is_contiguous() and r.shape[1] == D) for r in (reg_user_w, reg_item_w))):
            go = grad_out.reshape(-1).contiguous().to(torch.float32)
            uw, iw, gp, o4, gop = user_w.data_ptr(), item_w.data_ptr(), g.data_ptr(), out4.data_ptr(), go.data_ptr()
            sep_u = reg_user_w is not None and reg_user_w.data_ptr() != uw
            sep_i = reg_item_w is not None and reg_item_w.data_ptr() != iw
            if shared:
                gU = torch.zeros_like(user_w)
                gI = gU
            else:
                gU, gI = _zeros_like2(user_w, item_w)
            gRU = torch.zeros_like(reg_user_w) if reg_user_w is not None else None
            gRI = torch.zeros_like(reg_item_w) if reg_item_w is not None else None
            reg_kw = dict(r_stride=D, reg=ctx.reg_weight, B=n)
            su = _ord_seg(uid, coef=gp, go=gop, X=iw, xid=iid.data_ptr(), x_stride=D,
                          **({} if sep_u else dict(R=uw, norm=o4 + 8, **reg_kw)))
            si = _ord_seg(iid, coef=gp, go=gop, X=uw, xid=uid.data_ptr(), x_stride=D,
                          **({} if sep_i else dict(R=iw, norm=o4 + 12, **reg_kw)))
            lists = [(gU.data_ptr(), D, [su, si])] if shared else [(gU.data_ptr(), D, [su]), (gI.data_ptr(), D, [si])]
            if sep_u and ctx.reg_weight != 0.0:
                lists.append((gRU.data_ptr(), D, [_ord_seg(uid, go=gop, R=reg_user_w.data_ptr(), norm=o4 + 8, **reg_kw)]))
            if sep_i and ctx.reg_weight != 0.0:
                lists.append((gRI.data_ptr(), D, [_ord_seg(iid, go=gop, R=reg_item_w.data_ptr(), norm=o4 + 12, **reg_kw)]))
            _ordered_bwd(D, lists, keep=(go, gU, gI, gRU, gRI))
            return None, gU, (None if shared else gI), gRU, gRI, None, None, None, None
        if deterministic():
            go = grad_out.reshape(-1).contiguous().to(torch.float32)
            dU, dI, dRU, dRI = _det_point_rows(user_w, item_w, reg_user_w, reg_item_w, uid, iid, g, B_.f32(out4), ctx.reg_weight, go)
            if shared:
                gU = _scatter_rows_deterministic(user_w.shape, torch.cat([uid, iid]), torch.cat([dU, dI]))
                gI = None
            else:
                gU, gI = _scatter_rows_deterministic(user_w.shape, uid, dU), _scatter_rows_deterministic(item_w.shape, iid, dI)
            gRU = gRI = None
            if reg_user_w is not None:
                gRU = _scatter_rows_deterministic(reg_user_w.shape, uid, dRU) if dRU is not None else torch.zeros_like(reg_user_w)
            if reg_item_w is not None:
                gRI = _scatter_rows_deterministic(reg_item_w.shape, iid, dRI) if dRI is not None else torch.zeros_like(reg_item_w)
            return None, gU, gI, gRU, gRI, None, None, None, None
        if shared:
            gU = torch.zeros_like(user_w)
            gI = gU
        else:
            gU, gI = _zeros_like2(user_w, item_w)
        gRU = torch.zeros_like(reg_user_w) if reg_user_w is not None else None
        gRI = torch.zeros_like(reg_item_w) if reg_item_w is not None else None
        go = grad_out.reshape(-1).contiguous().to(torch.float32)
        B_.call('cdr_point_bwd_dense', B_.ctx(user_w.device), B_.stream(), B_.f32(user_w), B_.f32(item_w),
                B_.f32(reg_user_w), B_.f32(reg_item_w), user_w.shape[1], B_.i64(uid), B_.i64(iid), uid.numel(),
                B_.f32(g), B_.f32(out4), ctx.reg_weight, B_.f32(go), B_.f32(gU), B_.f32(gI), B_.f32(gRU), B_.f32(gRI))
        return None, gU, (None if shared else gI), gRU, gRI, None, None, None, None


_pair_w = {}


def _pair_weights(dev, alpha):
    """[alpha, 1 - alpha] on the device, created once per (device, alpha): a host-to-device copy is not allowed inside a hipGraph capture
    (the eager warm-up steps that precede a capture create it)."""
    key = (str(dev), alpha)
    if key not in _pair_w:
        _pair_w[key] = torch.tensor([alpha, 1.0 - alpha], device=dev, dtype=torch.float32)
    return _pair_w[key]


class TwoDomainPointLoss(Function):
    """alpha * L(source batch) + (1 - alpha) * L(target batch) for two pointwise batches on the SAME pair of tables (cmf.py:81-99: both
    domains share user_embedding / item_embedding): two fused gather-dot-loss launches, and in the backward two scatter launches
    into ONE pair of gradient buffers -- instead of two autograd nodes with their own table-sized gradients, the adds that merge
    them and the scalar arithmetic in between.  Returns (total [1], per-domain losses [2])."""

    @staticmethod
    def forward(ctx, kind, user_w, item_w, su, si, sl, reg_s, tu, ti, tl, reg_t, alpha):
        _dev_check(user_w, item_w, su, si, tu, ti)
        dev = user_w.device
        D = user_w.shape[1]
        ids = [_ids(su), _ids(si), _ids(tu), _ids(ti)]
        labels = [sl.reshape(-1).contiguous().to(torch.float32), tl.reshape(-1).contiguous().to(torch.float32)]
        out8 = torch.empty(2, 4, device=dev, dtype=torch.float32)
        w = _pair_weights(dev, float(alpha))
        total = torch.empty(1, device=dev, dtype=torch.float32)
        gs = [torch.empty(ids[0].numel(), device=dev, dtype=torch.float32), torch.empty(ids[2].numel(), device=dev, dtype=torch.float32)]
        ctx.pre = None
        if D % 4 == 0:
            # both batches in one gather-dot-loss launch and one finishing block that also forms alpha * L_s + (1 - alpha) * L_t
            _prezero_for_backward(ctx, user_w, item_w)
            P2, I2, F2 = ctypes.c_void_p * 2, ctypes.c_int64 * 2, ctypes.c_float * 2
            keep = [user_w, item_w, out8, w, total] + ids + labels + gs
            B_.call('cdr_point_fwd_pair', B_.ctx(dev), B_.stream(), int(kind), P2(user_w.data_ptr(), user_w.data_ptr()),
                    P2(item_w.data_ptr(), item_w.data_ptr()), None, None, D, P2(ids[0].data_ptr(), ids[2].data_ptr()),
                    P2(ids[1].data_ptr(), ids[3].data_ptr()), P2(labels[0].data_ptr(), labels[1].data_ptr()), I2(ids[0].numel(), ids[2].numel()),
                    F2(float(reg_s), float(reg_t)), P2(out8.data_ptr(), out8.data_ptr() + 16), P2(gs[0].data_ptr(), gs[1].data_ptr()), None,
                    B_.f32(w), B_.f32(total))
            del keep
        else:
            for d, (u, i, y, reg) in enumerate(((ids[0], ids[1], labels[0], reg_s), (ids[2], ids[3], labels[1], reg_t))):
                B_.call('cdr_point_fwd', B_.ctx(dev), B_.stream(), int(kind), B_.f32(user_w), B_.f32(item_w), None, None, D, B_.i64(u), B_.i64(i),
                        B_.f32(y), u.numel(), float(reg), B_._c_ptr(out8.data_ptr() + 16 * d), B_.f32(gs[d]), None)
            B_.call('cdr_scalar_mix', B_.stream(), 0, 2, B_.f32(out8), 4, B_.f32(w), None, B_.f32(total))      # (losses * w).sum(): one launch
        ctx.save_for_backward(user_w, item_w, *ids, *gs, out8, w)
        ctx.regs = (float(reg_s), float(reg_t))
        ctx.alpha = float(alpha)
        losses = out8[:, 0]
        ctx.mark_non_differentiable(losses)
        ctx.set_materialize_grads(False)      # no zero-filled gradient for the non-differentiable output (a launch per step)
        return total, losses

    @staticmethod
    def backward(ctx, grad_out, _gl):
        user_w, item_w, su, si, tu, ti, g_s, g_t, out8, w = ctx.saved_tensors
        dev, D = user_w.device, user_w.shape[1]
        go = grad_out.reshape(-1)[:1].contiguous().to(torch.float32)
        ns, nt = su.numel(), tu.numel()
        if _ord_fits(D, ns + nt) and user_w.is_contiguous() and item_w.is_contiguous():
            gU, gI = _grads_for(ctx, user_w, item_w)
            uw, iw, o8, gop = user_w.data_ptr(), item_w.data_ptr(), out8.data_ptr(), go.data_ptr()
            segs_u, segs_i = [], []
            for d, (u, i, gc, reg, n_, sc) in enumerate(((su, si, g_s, ctx.regs[0], ns, ctx.alpha), (tu, ti, g_t, ctx.regs[1], nt, 1.0 - ctx.alpha))):
                kw = dict(coef=gc.data_ptr(), go=gop, go_scale=sc, x_stride=D, r_stride=D, reg=reg, B=n_)
                segs_u.append(_ord_seg(u, X=iw, xid=i.data_ptr(), R=uw, norm=o8 + 16 * d + 8, **kw))
                segs_i.append(_ord_seg(i, X=uw, xid=u.data_ptr(), R=iw, norm=o8 + 16 * d + 12, **kw))
            _ordered_bwd(D, [(gU.data_ptr(), D, segs_u), (gI.data_ptr(), D, segs_i)], keep=(go, gU, gI))
            return None, gU, gI, None, None, None, None, None, None, None, None, None
        if deterministic():
            go2 = torch.empty(2, device=dev, dtype=torch.float32)
            B_.call('cdr_scalar_mix', B_.stream(), 1, 2, None, 0, B_.f32(w), B_.f32(go), B_.f32(go2))
            rows = [_det_point_rows(user_w, item_w, None, None, u, i, g, B_._c_ptr(out8.data_ptr() + 16 * d), reg, go2[d:d + 1])
                    for d, (u, i, g, reg) in enumerate(((su, si, g_s, ctx.regs[0]), (tu, ti, g_t, ctx.regs[1])))]
            gU = _scatter_rows_deterministic(user_w.shape, torch.cat([su, tu]), torch.cat([rows[0][0], rows[1][0]]))
            gI = _scatter_rows_deterministic(item_w.shape, torch.cat([si, ti]), torch.cat([rows[0][1], rows[1][1]]))
            return None, gU, gI, None, None, None, None, None, None, None, None, None
        gU, gI = _grads_for(ctx, user_w, item_w)
        if D % 4 == 0:
            # d total / d L_domain = grad_out * weight: the scatter kernel multiplies (the weights are host floats, as in _pair_weights)
            P2, I2, F2 = ctypes.c_void_p * 2, ctypes.c_int64 * 2, ctypes.c_float * 2
            B_.call('cdr_point_bwd_dense_pair', B_.ctx(dev), B_.stream(), P2(user_w.data_ptr(), user_w.data_ptr()), P2(item_w.data_ptr(), item_w.data_ptr()),
                    None, None, D, P2(su.data_ptr(), tu.data_ptr()), P2(si.data_ptr(), ti.data_ptr()), I2(su.numel(), tu.numel()),
                    P2(g_s.data_ptr(), g_t.data_ptr()), P2(out8.data_ptr(), out8.data_ptr() + 16), F2(*ctx.regs),
                    P2(go.data_ptr(), go.data_ptr()), F2(ctx.alpha, 1.0 - ctx.alpha), P2(gU.data_ptr(), gU.data_ptr()),
                    P2(gI.data_ptr(), gI.data_ptr()), None, None)
        else:
            go2 = torch.empty(2, device=dev, dtype=torch.float32)
            B_.call('cdr_scalar_mix', B_.stream(), 1, 2, None, 0, B_.f32(w), B_.f32(go), B_.f32(go2))
            for d, (u, i, g, reg) in enumerate(((su, si, g_s, ctx.regs[0]), (tu, ti, g_t, ctx.regs[1]))):
                B_.call('cdr_point_bwd_dense', B_.ctx(dev), B_.stream(), B_.f32(user_w), B_.f32(item_w), None, None, D,
                        B_.i64(u), B_.i64(i), u.numel(), B_.f32(g), B_._c_ptr(out8.data_ptr() + 16 * d), reg, B_._c_ptr(go2.data_ptr() + 4 * d),
                        B_.f32(gU), B_.f32(gI), None, None)
        return None, gU, gI, None, None, None, None, None, None, None, None, None


class TwoStackPointLoss(Function):
    """Two pointwise losses, each on its own stacked [users ; items] table (BiTGCF scores the batch rows of the two propagated stacks:
    bitgcf.py:222-240; item i is row ``nu + i`` of its stack -- the item operand is simply the stack from row ``nu`` on, so the ids go in
    unshifted): one gather-dot-loss launch and one finishing block
    for both, one scatter launch into two gradient buffers out of ONE zero-fill -- half the launches of two PointGatherLoss nodes.
    Returns (loss_s [1], loss_t [1])."""

    @staticmethod
    def forward(ctx, kind, S, T, nu, us, is_, ls, ut, it, lt):
        _dev_check(S, T, us, is_, ut, it)
        dev, D = S.device, S.shape[1]
        assert D % 4 == 0 and S.is_contiguous() and T.is_contiguous()
        ids = [_ids(us), _ids(is_), _ids(ut), _ids(it)]
        labels = [ls.reshape(-1).contiguous().to(torch.float32), lt.reshape(-1).contiguous().to(torch.float32)]
        out8 = torch.empty(2, 4, device=dev, dtype=torch.float32)
        gs = [torch.empty(ids[0].numel(), device=dev, dtype=torch.float32), torch.empty(ids[2].numel(), device=dev, dtype=torch.float32)]
        P2, I2, F2 = ctypes.c_void_p * 2, ctypes.c_int64 * 2, ctypes.c_float * 2
        io = 4 * int(nu) * D                                          # byte offset of the item rows inside a stack
        B_.call('cdr_point_fwd_pair', B_.ctx(dev), B_.stream(), int(kind), P2(S.data_ptr(), T.data_ptr()), P2(S.data_ptr() + io, T.data_ptr() + io), None, None,
                D, P2(ids[0].data_ptr(), ids[2].data_ptr()), P2(ids[1].data_ptr(), ids[3].data_ptr()), P2(labels[0].data_ptr(), labels[1].data_ptr()),
                I2(ids[0].numel(), ids[2].numel()), F2(0.0, 0.0), P2(out8.data_ptr(), out8.data_ptr() + 16), P2(gs[0].data_ptr(), gs[1].data_ptr()),
                None, None, None)
        ctx.save_for_backward(S, T, *ids, *gs, out8)
        ctx.nu = int(nu)
        ctx.set_materialize_grads(False)
        return out8[0, :1], out8[1, :1]

    @staticmethod
    def backward(ctx, g_s, g_t):
        S, T, us, is_, ut, it, gc_s, gc_t, out8 = ctx.saved_tensors
        dev, D = S.device, S.shape[1]
        zero = None
        gos = []
        for g in (g_s, g_t):
            if g is None:
                zero = torch.zeros(1, device=dev, dtype=torch.float32) if zero is None else zero
                g = zero
            gos.append(g.reshape(-1)[:1].contiguous().to(torch.float32))
        if _ord_fits(D, us.numel(), ut.numel()):
            gS, gT = _zeros_like2(S, T)
            io = 4 * ctx.nu * D
            lists = []
            for W, gW, u, i, gc, gop in ((S, gS, us, is_, gc_s, gos[0]), (T, gT, ut, it, gc_t, gos[1])):
                kw = dict(coef=gc.data_ptr(), go=gop.data_ptr(), x_stride=D)
                lists.append((gW.data_ptr(), D, [_ord_seg(u, X=W.data_ptr() + io, xid=i.data_ptr(), **kw)]))        # user rows < nu
                lists.append((gW.data_ptr() + io, D, [_ord_seg(i, X=W.data_ptr(), xid=u.data_ptr(), **kw)]))        # item rows from nu on
            _ordered_bwd(D, lists, keep=(gos, gS, gT))
            return None, gS, gT, None, None, None, None, None, None, None
        if deterministic():
            # per-occurrence rows of each stack (users, then items at row nu + i), then one in-order segment sum per stack
            outs = []
            for d, (W, u, i, gc) in enumerate(((S, us, is_, gc_s), (T, ut, it, gc_t))):
                Wi = W[ctx.nu:]
                dU, dI, _, _ = _det_point_rows(W, Wi, None, None, u, i, gc, B_._c_ptr(out8.data_ptr() + 16 * d), 0.0, gos[d])
                outs.append(_scatter_rows_deterministic(W.shape, torch.cat([u, i + ctx.nu]), torch.cat([dU, dI])))
            return None, outs[0], outs[1], None, None, None, None, None, None, None
        gS, gT = _zeros_like2(S, T)
        P2, I2, F2 = ctypes.c_void_p * 2, ctypes.c_int64 * 2, ctypes.c_float * 2
        io = 4 * ctx.nu * D
        B_.call('cdr_point_bwd_dense_pair', B_.ctx(dev), B_.stream(), P2(S.data_ptr(), T.data_ptr()), P2(S.data_ptr() + io, T.data_ptr() + io), None, None, D,
                P2(us.data_ptr(), ut.data_ptr()), P2(is_.data_ptr(), it.data_ptr()), I2(us.numel(), ut.numel()),
                P2(gc_s.data_ptr(), gc_t.data_ptr()), P2(out8.data_ptr(), out8.data_ptr() + 16), F2(0.0, 0.0),
                P2(gos[0].data_ptr(), gos[1].data_ptr()), None, P2(gS.data_ptr(), gT.data_ptr()), P2(gS.data_ptr() + io, gT.data_ptr() + io), None, None)
        del gos
        return None, gS, gT, None, None, None, None, None, None, None


class TwoPointLoss(Function):
    """Two pointwise losses on two different (user, item) table pairs in one launch each way (CLFM: each domain's factor rows against its
    item table, clfm.py:87-113) -- PointGatherLoss twice, with half the launches.  Returns (loss_0 [1], loss_1 [1], scores_0, scores_1)."""

    @staticmethod
    def forward(ctx, kind, U0, I0, U1, I1, u0, i0, l0, u1, i1, l1):
        _dev_check(U0, I0, U1, I1, u0, i0, u1, i1)
        dev, D = U0.device, U0.shape[1]
        assert D % 4 == 0 and I0.shape[1] == D and U1.shape[1] == D and I1.shape[1] == D
        tabs = [t.contiguous() for t in (U0, I0, U1, I1)]
        ids = [_ids(u0), _ids(i0), _ids(u1), _ids(i1)]
        labels = [l0.reshape(-1).contiguous().to(torch.float32), l1.reshape(-1).contiguous().to(torch.float32)]
        n0, n1 = ids[0].numel(), ids[2].numel()
        out8 = torch.empty(2, 4, device=dev, dtype=torch.float32)
        gs = [torch.empty(n0, device=dev, dtype=torch.float32), torch.empty(n1, device=dev, dtype=torch.float32)]
        sc = [torch.empty(n0, device=dev, dtype=torch.float32), torch.empty(n1, device=dev, dtype=torch.float32)]
        P2, I2, F2 = ctypes.c_void_p * 2, ctypes.c_int64 * 2, ctypes.c_float * 2
        B_.call('cdr_point_fwd_pair', B_.ctx(dev), B_.stream(), int(kind), P2(tabs[0].data_ptr(), tabs[2].data_ptr()), P2(tabs[1].data_ptr(), tabs[3].data_ptr()),
                None, None, D, P2(ids[0].data_ptr(), ids[2].data_ptr()), P2(ids[1].data_ptr(), ids[3].data_ptr()),
                P2(labels[0].data_ptr(), labels[1].data_ptr()), I2(n0, n1), F2(0.0, 0.0), P2(out8.data_ptr(), out8.data_ptr() + 16),
                P2(gs[0].data_ptr(), gs[1].data_ptr()), P2(sc[0].data_ptr(), sc[1].data_ptr()), None, None)
        ctx.save_for_backward(*tabs, *ids, *gs, out8)
        ctx.mark_non_differentiable(sc[0], sc[1])
        ctx.set_materialize_grads(False)
        return out8[0, :1], out8[1, :1], sc[0], sc[1]

    @staticmethod
    def backward(ctx, g0, g1, _s0, _s1):
        U0, I0, U1, I1, u0, i0, u1, i1, gc0, gc1, out8 = ctx.saved_tensors
        dev, D = U0.device, U0.shape[1]
        sizes = [t.numel() for t in (U0, I0, U1, I1)]
        flat = torch.zeros(sum(sizes), device=dev, dtype=torch.float32)            # four gradient tables, one fill
        gt, o = [], 0
        for t, n in zip((U0, I0, U1, I1), sizes):
            gt.append(flat[o:o + n].view(t.shape)); o += n
        zero = None
        gos = []
        for g in (g0, g1):
            if g is None:
                zero = torch.zeros(1, device=dev, dtype=torch.float32) if zero is None else zero
                g = zero
            gos.append(g.reshape(-1)[:1].contiguous().to(torch.float32))
        P2, I2, F2 = ctypes.c_void_p * 2, ctypes.c_int64 * 2, ctypes.c_float * 2
        B_.call('cdr_point_bwd_dense_pair', B_.ctx(dev), B_.stream(), P2(U0.data_ptr(), U1.data_ptr()), P2(I0.data_ptr(), I1.data_ptr()), None, None, D,
                P2(u0.data_ptr(), u1.data_ptr()), P2(i0.data_ptr(), i1.data_ptr()), I2(u0.numel(), u1.numel()), P2(gc0.data_ptr(), gc1.data_ptr()),
                P2(out8.data_ptr(), out8.data_ptr() + 16), F2(0.0, 0.0), P2(gos[0].data_ptr(), gos[1].data_ptr()), None,
                P2(gt[0].data_ptr(), gt[2].data_ptr()), P2(gt[1].data_ptr(), gt[3].data_ptr()), None, None)
        del gos
        return None, gt[0], gt[1], gt[2], gt[3], None, None, None, None, None, None


class GatherRows(Function):
    """nn.Embedding(idx) with its dense backward."""

    @staticmethod
    def forward(ctx, weight, ids):
        _dev_check(weight, ids)
        shape = tuple(ids.shape)
        flat = _ids(ids)
        D = weight.shape[1]
        out = torch.empty(flat.numel(), D, device=weight.device, dtype=torch.float32)
        if flat.numel():
            B_.call('cdr_gather_rows', B_.stream(), B_.f32(weight), D, B_.i64(flat), flat.numel(), B_.f32(out))
        ctx.save_for_backward(flat)
        ctx.wshape = tuple(weight.shape)
        return out.view(*shape, D)

    @staticmethod
    def backward(ctx, grad_out):
        (flat,) = ctx.saved_tensors
        if _ord_fits(ctx.wshape[1], flat.numel()):
            D = ctx.wshape[1]
            gW = torch.zeros(ctx.wshape, device=grad_out.device, dtype=torch.float32)
            go = grad_out.reshape(-1, D).contiguous().to(torch.float32)
            _ordered_bwd(D, [(gW.data_ptr(), D, [_ord_seg(flat, X=go.data_ptr(), x_stride=D)])], keep=(go, gW))
            return gW, None
        if deterministic():
            return _scatter_rows_deterministic(ctx.wshape, flat, grad_out.reshape(-1, ctx.wshape[1])), None
        gW = torch.zeros(ctx.wshape, device=grad_out.device, dtype=torch.float32)
        if flat.numel():
            go = grad_out.reshape(-1, ctx.wshape[1]).contiguous()
            B_.call('cdr_scatter_add_rows', B_.stream(), B_.f32(gW), ctx.wshape[1], B_.i64(flat), flat.numel(),
                    B_.f32(go), None)
        return gW, None


def gather_rows(weight, ids):
    return GatherRows.apply(weight, ids)


def _ld(t):
    """Leading dimension of a 2-D operand whose rows are unit-stride (column slices of a wider matrix are fine)."""
    if t.dim() != 2 or t.stride(1) != 1:
        raise ValueError('gemm operands must be 2-D with unit column stride')
    return t.stride(0) if t.shape[0] > 1 else max(t.stride(0), t.shape[1])


def gemm(a, b, trans_a=False, trans_b=False, bias=None, act=B_.ACT_NONE, out=None, accumulate=0, rowscale=None):
    """C = epi(op(a) @ op(b)) on the fp32 MFMA kernel.  accumulate: 0 act(v+bias) ; 1 act(v+bias)+C ; 2 act((C+v)+bias),
    v = rowscale[m] * acc."""
    _dev_check(a, b, bias, rowscale)
    M = a.shape[1] if trans_a else a.shape[0]
    K = a.shape[0] if trans_a else a.shape[1]
    N = b.shape[0] if trans_b else b.shape[1]
    Kb = b.shape[1] if trans_b else b.shape[0]
    if K != Kb:
        raise ValueError(f'gemm: inner dimensions differ ({K} vs {Kb})')
    if out is None:
        out = torch.empty(M, N, device=a.device, dtype=torch.float32)
    B_.call('cdr_gemm_f32_ex', B_.stream(), int(trans_a), int(trans_b), M, N, K, B_._c_ptr(a.data_ptr()), _ld(a),
            B_._c_ptr(b.data_ptr()), _ld(b), B_._c_ptr(out.data_ptr()), _ld(out), B_.f32(bias), B_.f32(rowscale), int(act),
            int(accumulate))
    return out


_WGRAD_CHUNKED_ROWS = 1 << 18  # ... and up to here the weight gradients alone (64 chunks of <= 4,096 rows per tile)
_SMALL_WGRAD_ROWS = 16384    # batches up to here take the one-wave-per-tile kernels of csrc/cdr_linear.hip (dW: 512-row chunks per workgroup)


def _wgrad_tiles_fit(dout, din):
    """cdr_linear_wgrad_small hands one ticket to every 32 x 32 tile of dW and has CDR_TICKETS = 1024 of them (csrc/cdr_linear.hip)."""
    return (dout + 31) // 32 * ((din + 31) // 32) <= 1024


def _small_linear(x2, weight):
    """Small batches go to the one-wave-per-tile kernels of csrc/cdr_linear.hip (the general contraction's start-up dominates them)."""
    return (x2.shape[0] <= _SMALL_WGRAD_ROWS and x2.shape[1] % 4 == 0 and x2.data_ptr() % 16 == 0 and weight.data_ptr() % 16 == 0
            and weight.is_contiguous())


class LinearAct(Function):
    """y = act(x W^T + b)  (nn.Linear + Tanh/ReLU/Sigmoid: emcdr.py:86-93, conet.py:74-84, recbole MLPLayers)."""

    @staticmethod
    def forward(ctx, x, weight, bias, act):
        shape = tuple(x.shape)
        x2 = x.reshape(-1, shape[-1]).contiguous()
        if _small_linear(x2, weight):
            w_ = weight.contiguous()
            y = torch.empty(x2.shape[0], w_.shape[0], device=x2.device, dtype=torch.float32)
            B_.call('cdr_linear_small', B_.stream(), 0, B_.f32(x2), x2.shape[1], B_.f32(w_), w_.shape[1], x2.shape[0], w_.shape[0], w_.shape[1],
                    None if bias is None else B_.f32(bias.contiguous()), int(act), B_.f32(y), y.shape[1], None, 0)
        else:
            y = gemm(x2, weight.contiguous(), trans_b=True, bias=bias, act=act)
        ctx.save_for_backward(x2, weight, y)
        ctx.act, ctx.has_bias, ctx.xshape = act, bias is not None, shape
        return y.view(*shape[:-1], weight.shape[0])

    @staticmethod
    def backward(ctx, gy):
        x2, weight, y = ctx.saved_tensors
        gy2 = gy.reshape(-1, weight.shape[0]).contiguous()
        want_b = ctx.has_bias and ctx.needs_input_grad[2]
        w_ = weight.contiguous()
        rows, dout, din = gy2.shape[0], w_.shape[0], w_.shape[1]
        if rows <= _SMALL_WGRAD_ROWS and y.is_contiguous():
            # small batches (csrc/cdr_linear.hip): dx in one launch, dW + db in one launch, the activation's backward inside both --
            # instead of an activation pass, two contractions and a two-launch column sum
            gx = gW = gb = None
            dx_small = dout % 4 == 0 and gy2.data_ptr() % 16 == 0 and y.data_ptr() % 16 == 0
            gz, yp = gy2, (B_.f32(y) if ctx.act != B_.ACT_NONE else None)
            if ctx.needs_input_grad[0]:
                if dx_small:
                    gx = torch.empty(rows, din, device=gy2.device, dtype=torch.float32)
                    B_.call('cdr_linear_small', B_.stream(), 1, B_.f32(gy2), dout, B_.f32(w_), din, rows, din, dout, None, B_.ACT_NONE,
                            B_.f32(gx), din, yp, int(ctx.act))
                else:                                # widths that are not multiples of 4: gz by its own launch, dx on the general contraction
                    if ctx.act != B_.ACT_NONE:
                        gz = torch.empty_like(gy2)
                        B_.call('cdr_act_bwd', B_.stream(), ctx.act, B_.f32(y), B_.f32(gy2), B_.f32(gz), gy2.numel())
                        yp = None
                    gx = gemm(gz, w_)
                gx = gx.view(ctx.xshape)
            if (ctx.needs_input_grad[1] or want_b) and not _wgrad_tiles_fit(dout, din):
                # more 32 x 32 tiles than the one-launch kernel has tickets for (e.g. a 2048 x 1024 layer): the general route
                if yp is not None:
                    gz = torch.empty_like(gy2)
                    B_.call('cdr_act_bwd', B_.stream(), ctx.act, B_.f32(y), B_.f32(gy2), B_.f32(gz), gy2.numel())
                if ctx.needs_input_grad[1]:
                    gW = gemm(gz, x2, trans_a=True)
                if want_b:
                    gb = torch.empty(dout, device=gy2.device, dtype=torch.float32)
                    B_.call('cdr_colsum', B_.ctx(gy2.device), B_.stream(), B_.f32(gz), rows, dout, B_.f32(gb), 0)
            elif ctx.needs_input_grad[1] or want_b:
                gW = torch.empty_like(w_)
                gb = torch.empty(dout, device=gy2.device, dtype=torch.float32) if want_b else None
                need = ctypes.c_size_t(0)
                B_._check(B_.load().cdr_linear_wgrad_small_workspace(rows, dout, din, ctypes.byref(need)), 'cdr_linear_wgrad_small_workspace')
                # (a tensor of this call's own: under a hipGraph capture it comes from the graph's pool and stays where the graph expects it)
                ws = torch.empty(int(need.value), device=gy2.device, dtype=torch.uint8) if need.value else None
                B_.call('cdr_linear_wgrad_small', B_.ctx(gy2.device), B_.stream(), B_.f32(gz), yp, int(ctx.act), B_.f32(x2), rows, dout, din,
                        B_.f32(gW), None if gb is None else B_.f32(gb), None if ws is None else B_.raw(ws), int(need.value))
                if not ctx.needs_input_grad[1]:
                    gW = None
            return gx, gW, gb, None
        if ctx.act != B_.ACT_NONE:
            gz = torch.empty_like(gy2)
            B_.call('cdr_act_bwd', B_.stream(), ctx.act, B_.f32(y), B_.f32(gy2), B_.f32(gz), gy2.numel())
        else:
            gz = gy2
        gx = gW = gb = None
        if ctx.needs_input_grad[0]:
            gx = gemm(gz, w_).view(ctx.xshape)                                      # [rows,out] x [out,in]
        if (ctx.needs_input_grad[1] or want_b) and rows <= _WGRAD_CHUNKED_ROWS and _wgrad_tiles_fit(dout, din):
            # larger batches: dx stays on the general contraction, dW + db still come from ONE fixed-order launch (up to 64 row chunks per
            # tile) instead of a transposed GEMM with split-K atomics + a two-launch column sum
            gW = torch.empty_like(w_)
            gb = torch.empty(dout, device=gy2.device, dtype=torch.float32) if want_b else None
            need = ctypes.c_size_t(0)
            B_._check(B_.load().cdr_linear_wgrad_small_workspace(rows, dout, din, ctypes.byref(need)), 'cdr_linear_wgrad_small_workspace')
            ws = torch.empty(int(need.value), device=gy2.device, dtype=torch.uint8) if need.value else None
            B_.call('cdr_linear_wgrad_small', B_.ctx(gy2.device), B_.stream(), B_.f32(gz), None, 0, B_.f32(x2), rows, dout, din,
                    B_.f32(gW), None if gb is None else B_.f32(gb), None if ws is None else B_.raw(ws), int(need.value))
            return gx, (gW if ctx.needs_input_grad[1] else None), gb, None
        if ctx.needs_input_grad[1]:
            gW = gemm(gz, x2, trans_a=True)                                         # [out,rows] x [rows,in]
        if want_b:
            gb = torch.empty(weight.shape[0], device=gy.device, dtype=torch.float32)
            B_.call('cdr_colsum', B_.ctx(gy.device), B_.stream(), B_.f32(gz), gz.shape[0], gz.shape[1], B_.f32(gb), 0)
        return gx, gW, gb, None


def linear(x, weight, bias=None, act=B_.ACT_NONE):
    return LinearAct.apply(x, weight, bias, act)


class MSELoss(Function):
    """nn.MSELoss(): mean over all elements (emcdr.py:81,162,167)."""

    @staticmethod
    def forward(ctx, a, b):
        _dev_check(a, b)
        a_, b_ = a.contiguous(), b.contiguous()
        out = torch.empty(1, device=a.device, dtype=torch.float32)
        B_.call('cdr_mse_fwd', B_.ctx(a.device), B_.stream(), B_.f32(a_), B_.f32(b_), a_.numel(), B_.f32(out))
        ctx.save_for_backward(a_, b_)
        return out.reshape(())

    @staticmethod
    def backward(ctx, grad_out):
        a_, b_ = ctx.saved_tensors
        ga = torch.empty_like(a_) if ctx.needs_input_grad[0] else None
        gb = torch.empty_like(b_) if ctx.needs_input_grad[1] else None
        go = grad_out.reshape(-1).contiguous()
        B_.call('cdr_mse_bwd', B_.stream(), B_.f32(a_), B_.f32(b_), a_.numel(), B_.f32(go), B_.f32(ga), B_.f32(gb))
        return ga, gb


def mse_loss(a, b):
    return MSELoss.apply(a, b)


def select_mapped(mapped, weight, ids, n_overlap):
    """where(id < n_overlap, mapped, weight[id])  -- evaluation only (emcdr.py:195-197,222-224)."""
    _dev_check(mapped, weight, ids)
    flat = _ids(ids)
    D = weight.shape[1]
    out = torch.empty(flat.numel(), D, device=weight.device, dtype=torch.float32)
    B_.call('cdr_select_mapped', B_.stream(), B_.f32(mapped.detach().contiguous()), B_.f32(weight.detach()), D, B_.i64(flat),
            flat.numel(), int(n_overlap), B_.f32(out))
    return out


def fullsort_scores(user_e, slab0, slab1=None, out=None):
    """scores[U, n0+n1] = user_e @ cat(slab0, slab1)^T without the cat copy; slabs are contiguous row ranges."""
    _dev_check(user_e, slab0, slab1)
    user_e = user_e.detach().contiguous()
    U, D = user_e.shape
    n0 = slab0.shape[0] if slab0 is not None else 0
    n1 = slab1.shape[0] if slab1 is not None else 0
    if out is None:
        out = torch.empty(U, n0 + n1, device=user_e.device, dtype=torch.float32)
    assert out.is_contiguous() and tuple(out.shape) == (U, n0 + n1)
    B_.call('cdr_fullsort_scores_f32', B_.stream(), B_.f32(user_e), U, D,
            B_.f32(slab0.detach()) if n0 else None, n0, B_.f32(slab1.detach()) if n1 else None, n1, B_.f32(out))
    return out


_topk_ws = {}


def fullsort_topk(user_e, slab0, slab1=None, k=10, hist_indptr=None, hist_cols=None, exclude_first_col=True):
    """(values [U,k] descending, columns [U,k] int64) of ``fullsort_scores(user_e, slab0, slab1)`` after the evaluation
    mask -- column 0 (PAD) and, per user, the ascending columns ``hist_cols[hist_indptr[u]:hist_indptr[u+1]]`` -- without
    materialising the [U, N] matrix when U > 32 and D is 64 or 128 (cdr_fullsort_topk_f32)."""
    _dev_check(user_e, slab0, slab1, hist_indptr, hist_cols)
    user_e = user_e.detach().contiguous()
    U, D = user_e.shape
    n0 = slab0.shape[0] if slab0 is not None else 0
    n1 = slab1.shape[0] if slab1 is not None else 0
    dev = user_e.device
    need = ctypes.c_size_t(0)
    B_._check(B_.load().cdr_fullsort_topk_workspace_bytes(U, D, n0, n1, int(k), ctypes.byref(need)),
              'cdr_fullsort_topk_workspace_bytes')
    key = (dev.index, torch.cuda.current_stream(dev).cuda_stream)
    ws = _topk_ws.get(key)
    if ws is None or ws.numel() < need.value:
        ws = torch.empty(int(need.value), device=dev, dtype=torch.uint8)
        _topk_ws[key] = ws
    vals = torch.empty(U, k, device=dev, dtype=torch.float32)
    idx = torch.empty(U, k, device=dev, dtype=torch.int64)
    B_.call('cdr_fullsort_topk_f32', B_.stream(), B_.f32(user_e), U, D, B_.f32(slab0.detach()) if n0 else None, n0,
            B_.f32(slab1.detach()) if n1 else None, n1, int(k), B_.i64(hist_indptr), B_.i64(hist_cols),
            1 if exclude_first_col else 0, B_.f32(vals), B_.i64(idx), B_.raw(ws), ws.numel())
    return vals, idx


def fullsort_neg_sqdist(user_e, items):
    _dev_check(user_e, items)
    user_e, items = user_e.detach().contiguous(), items.detach().contiguous()
    U, D = user_e.shape
    N = items.shape[0]
    out = torch.empty(U, N, device=user_e.device, dtype=torch.float32)
    scratch = torch.empty(U + N, device=user_e.device, dtype=torch.float32)
    B_.call('cdr_fullsort_neg_sqdist_f32', B_.stream(), B_.f32(user_e), U, D, B_.f32(items), N, B_.f32(scratch), B_.f32(out))
    return out


# ---------------------------------------------------------------------------------------------------- CoNet pieces
class GatherConcat2(Function):
    """cat([A[ida], B[idb]], dim=1) in one buffer (conet.py:106-111) with the dense scatter-add backward."""

    @staticmethod
    def forward(ctx, wa, wb, ida, idb):
        _dev_check(wa, wb, ida, idb)
        ida, idb = _ids(ida), _ids(idb)
        n, D = ida.numel(), wa.shape[1]
        out = torch.empty(n, 2 * D, device=wa.device, dtype=torch.float32)
        B_.call('cdr_gather_rows_ld', B_.stream(), B_.f32(wa), D, B_.i64(ida), n, B_.f32(out), 2 * D)
        B_.call('cdr_gather_rows_ld', B_.stream(), B_.f32(wb), D, B_.i64(idb), n, B_._c_ptr(out.data_ptr() + 4 * D), 2 * D)
        ctx.save_for_backward(ida, idb)
        ctx.shapes = (tuple(wa.shape), tuple(wb.shape))
        return out

    @staticmethod
    def backward(ctx, g):
        ida, idb = ctx.saved_tensors
        g = g.contiguous()
        D = ctx.shapes[0][1]
        ga = torch.zeros(ctx.shapes[0], device=g.device, dtype=torch.float32)
        gb = torch.zeros(ctx.shapes[1], device=g.device, dtype=torch.float32)
        B_.call('cdr_scatter_add_rows_ld', B_.stream(), B_.f32(ga), D, B_.i64(ida), ida.numel(), B_.f32(g), 2 * D)
        B_.call('cdr_scatter_add_rows_ld', B_.stream(), B_.f32(gb), D, B_.i64(idb), idb.numel(),
                B_._c_ptr(g.data_ptr() + 4 * D), 2 * D)
        return ga, gb, None, None


def overlap_mask(ids, n_overlap):
    ids = _ids(ids)
    out = torch.empty(ids.numel(), device=ids.device, dtype=torch.float32)
    B_.call('cdr_overlap_mask', B_.stream(), B_.i64(ids), ids.numel(), int(n_overlap), B_.f32(out))
    return out


def _rowscale(x, scale):
    out = torch.empty_like(x)
    B_.call('cdr_rowscale', B_.stream(), B_.f32(x), B_.f32(scale), x.shape[0], x.shape[1], B_.f32(out))
    return out


def _act_bwd(act, y, gy):
    gz = torch.empty_like(gy)
    B_.call('cdr_act_bwd', B_.stream(), act, B_.f32(y), B_.f32(gy), B_.f32(gz), gy.numel())
    return gz


def _colsum(x):
    out = torch.empty(x.shape[1], device=x.device, dtype=torch.float32)
    B_.call('cdr_colsum', B_.ctx(x.device), B_.stream(), B_.f32(x), x.shape[0], x.shape[1], B_.f32(out), 0)
    return out


class CrossUnit(Function):
    """One CoNet cross-connection layer for both towers (conet.py:118-137):
         s' = relu(s Ws^T + bs + m (.) (t H^T)),  t' = relu(t Wt^T + bt + m (.) (s H^T)),  m = 1 on overlapped rows."""

    @staticmethod
    def forward(ctx, s, t, Ws, bs, Wt, bt, H, m):
        s, t = s.contiguous(), t.contiguous()
        so = gemm(s, Ws, trans_b=True, bias=bs)
        gemm(t, H, trans_b=True, out=so, accumulate=2, rowscale=m, act=B_.ACT_RELU)
        to = gemm(t, Wt, trans_b=True, bias=bt)
        gemm(s, H, trans_b=True, out=to, accumulate=2, rowscale=m, act=B_.ACT_RELU)
        ctx.save_for_backward(s, t, Ws, Wt, H, m, so, to)
        return so, to

    @staticmethod
    def backward(ctx, gso, gto):
        s, t, Ws, Wt, H, m, so, to = ctx.saved_tensors
        gzs = _act_bwd(B_.ACT_RELU, so, gso.contiguous())
        gzt = _act_bwd(B_.ACT_RELU, to, gto.contiguous())
        mgzs, mgzt = _rowscale(gzs, m), _rowscale(gzt, m)
        gWs = gemm(gzs, s, trans_a=True)
        gWt = gemm(gzt, t, trans_a=True)
        gH = gemm(mgzs, t, trans_a=True)
        gemm(mgzt, s, trans_a=True, out=gH, accumulate=1)
        gs = gemm(gzs, Ws)
        gemm(mgzt, H, out=gs, accumulate=1)
        gt = gemm(gzt, Wt)
        gemm(mgzs, H, out=gt, accumulate=1)
        return gs, gt, gWs, _colsum(gzs), gWt, _colsum(gzt), gH, None


class BCEProbLoss(Function):
    """nn.BCELoss() on probabilities (conet.py:63,195-196)."""

    @staticmethod
    def forward(ctx, p, y):
        p_, y_ = p.reshape(-1).contiguous(), y.reshape(-1).contiguous().to(torch.float32)
        out = torch.empty(1, device=p.device, dtype=torch.float32)
        B_.call('cdr_bce_prob_fwd', B_.ctx(p.device), B_.stream(), B_.f32(p_), B_.f32(y_), p_.numel(), B_.f32(out))
        ctx.save_for_backward(p_, y_)
        ctx.pshape = tuple(p.shape)
        return out.reshape(())

    @staticmethod
    def backward(ctx, go):
        p_, y_ = ctx.saved_tensors
        gp = torch.empty_like(p_)
        B_.call('cdr_bce_prob_bwd', B_.stream(), B_.f32(p_), B_.f32(y_), p_.numel(), B_.f32(go.reshape(-1).contiguous()), B_.f32(gp))
        return gp.view(ctx.pshape), None


class FrobeniusNorm(Function):
    """torch.norm(W) (conet.py:198-201)."""

    @staticmethod
    def forward(ctx, w):
        w_ = w.contiguous()
        out = torch.empty(1, device=w.device, dtype=torch.float32)
        B_.call('cdr_frobenius_fwd', B_.ctx(w.device), B_.stream(), B_.f32(w_), w_.numel(), B_.f32(out))
        ctx.save_for_backward(w_, out)
        return out.reshape(()).clone()

    @staticmethod
    def backward(ctx, go):
        w_, norm = ctx.saved_tensors
        gw = torch.empty_like(w_)
        B_.call('cdr_frobenius_bwd', B_.stream(), B_.f32(w_), w_.numel(), B_.f32(norm), B_.f32(go.reshape(-1).contiguous()),
                B_.f32(gw), 0)
        return gw


_conet_ws = {}


def _conet_workspace(dev, need):
    """The tower kernels' scratch (weight-gradient partials, output-unit partials), one per device and stream."""
    key = (dev.index, torch.cuda.current_stream(dev).cuda_stream)
    ws = _conet_ws.get(key)
    if ws is None or ws.numel() < need:
        ws = torch.empty(need, device=dev, dtype=torch.uint8)
        _conet_ws[key] = ws
    return ws


_sort_bufs = {}


def sort_id_lists(lists, max_id):
    """[(keys_sorted, perm, n)] per id list (uint32 views of shared buffers): the chip-wide rank sort for lists <= 16384 ids,
    the radix sort above.  Buffers are cached per (device, stream, sizes)."""
    lists = [x.reshape(-1).contiguous().to(torch.int64) for x in lists]
    dev = lists[0].device
    ns = [int(x.numel()) for x in lists]
    key = (dev.index, torch.cuda.current_stream(dev).cuda_stream, tuple(ns))
    if key not in _sort_bufs:
        tot = sum(ns)
        _sort_bufs[key] = [torch.empty(tot, device=dev, dtype=torch.int32), torch.empty(tot, device=dev, dtype=torch.int32),
                           torch.zeros(tot, device=dev, dtype=torch.int32), None]
    keys, perm, rank, ws = _sort_bufs[key]
    offs, o = [], 0
    for n in ns:
        offs.append(o); o += n
    m = len(ns)
    if max(ns) <= 16384 and m <= 4:
        B_._alive.extend(lists)
        B_.call('cdr_sort_ids_small', B_.stream(), m, (ctypes.c_void_p * m)(*[x.data_ptr() for x in lists]), (ctypes.c_int64 * m)(*ns),
                None, None, (ctypes.c_int64 * m)(*offs), B_.raw(keys), B_.raw(perm), B_.raw(rank), int(max_id))
    else:
        for x, n, of in zip(lists, ns, offs):
            need = ctypes.c_size_t(0)
            B_._check(B_.load().cdr_sort_workspace_bytes(n, int(max_id), ctypes.byref(need)), 'cdr_sort_workspace_bytes')
            if ws is None or ws.numel() < need.value:
                ws = torch.empty(int(need.value), device=dev, dtype=torch.uint8)
                _sort_bufs[key][3] = ws
            B_.call('cdr_sort_ids', B_.ctx(dev), B_.stream(), B_.i64(x), n, None, 0, int(max_id), B_.raw(keys[of:of + n]),
                    B_.raw(perm[of:of + n]), B_.raw(ws), ws.numel())
    return [(keys[of:of + n], perm[of:of + n], n) for n, of in zip(ns, offs)]


def conet_supported(dims):
    """True when the fused tower kernels (csrc/cdr_conet.hip) take these layer widths ({2D, mlp_hidden_size...})."""
    aw, need = ctypes.c_int(0), ctypes.c_size_t(0)
    arr = (ctypes.c_int * len(dims))(*[int(d) for d in dims])
    return B_.load().cdr_conet_plan(len(dims) - 1, arr, 1024, ctypes.byref(aw), ctypes.byref(need)) == 0


def conet_fullsort_supported(h1, tail_dims):
    """True when csrc/cdr_conet_fullsort.hip takes a target tower of these widths behind the separable first layer."""
    arr = (ctypes.c_int * max(len(tail_dims), 1))(*[int(d) for d in tail_dims])
    return len(tail_dims) >= 1 and B_.load().cdr_conet_fullsort_supported(int(h1), len(tail_dims), arr) == 1


@torch.no_grad()
def conet_fullsort(P, Q, weights, biases, wo, bo, out=None):
    """[U, N] scores of CoNet.full_sort_predict (conet.py:222-242) from the two halves of the separable first layer -- P [N, h1]
    (items), Q [U, h1] (users + bias) -- the remaining layers ``weights[t]`` [d_t, d_in] / ``biases[t]`` and the output unit
    ``wo`` [d_last] / ``bo`` [1]: one launch, no intermediate in memory (csrc/cdr_conet_fullsort.hip)."""
    _dev_check(P, Q, wo, bo, *weights, *biases)
    U, N, h1 = Q.shape[0], P.shape[0], P.shape[1]
    assert Q.shape[1] == h1 and P.stride(1) == 1 and Q.stride(1) == 1
    T = len(weights)
    ws = [w.contiguous() for w in weights]
    bs = [b.contiguous() for b in biases]
    wo_, bo_ = wo.reshape(-1).contiguous(), bo.reshape(-1).contiguous()
    if out is None:
        out = torch.empty(U, N, device=P.device, dtype=torch.float32)
    dims = (ctypes.c_int * T)(*[int(w.shape[0]) for w in ws])
    B_._alive.extend(ws + bs + [P, Q])
    B_.call('cdr_conet_fullsort', B_.stream(), B_._c_ptr(P.data_ptr()), P.stride(0), B_._c_ptr(Q.data_ptr()), Q.stride(0), U, N, h1, T, dims,
            (ctypes.c_void_p * T)(*[w.data_ptr() for w in ws]), (ctypes.c_void_p * T)(*[b.data_ptr() for b in bs]), B_.f32(wo_), B_.f32(bo_),
            B_.f32(out), out.stride(0))
    return out


class ConetFullsortFewUsers:
    """CoNet.full_sort_predict for a FEW users as one launch (cdr_conet_fullsort_users): everything that does not change between calls --
    P, the tower, the table and first-layer pointers -- is packed into ctypes arguments ONCE; a call allocates the [U, N] scores and fills in
    the ids.  Built by the model in evaluation mode and dropped when it trains again (the pointers are those of the tensors held here)."""
    MAX_USERS = 8
    MAX_PAIRS = 262144       # beyond ~this many (user, item) pairs the scoring launch outlasts the host's three enqueues: nothing left to save

    @torch.no_grad()
    def __init__(self, P, user_table, W1, b1, D, weights, biases, wo, bo):
        _dev_check(P, user_table, W1, b1, wo, bo, *weights, *biases)
        assert P.stride(1) == 1 and user_table.stride(1) == 1 and W1.stride(1) == 1 and user_table.shape[1] >= D and W1.shape[1] >= D
        T = len(weights)
        self.keep = [P, user_table, W1, b1.contiguous(), [w.contiguous() for w in weights], [b.contiguous() for b in biases],
                     wo.reshape(-1).contiguous(), bo.reshape(-1).contiguous()]
        _P, _ut, _W1, _b1, ws, bs, wo_, bo_ = self.keep
        self.N, self.dev, self.table_ptr = P.shape[0], P.device, user_table.data_ptr()
        self.param_ptrs = tuple(t.data_ptr() for t in [W1, b1, wo, bo] + list(weights) + list(biases))      # (fresh(): the model checks them per call)
        self.dims = (ctypes.c_int * T)(*[int(w.shape[0]) for w in ws])
        self.Wp = (ctypes.c_void_p * T)(*[w.data_ptr() for w in ws])
        self.bp = (ctypes.c_void_p * T)(*[b.data_ptr() for b in bs])
        c = B_._c_ptr
        self.head = (c(P.data_ptr()), P.stride(0), c(user_table.data_ptr()), user_table.stride(0))
        self.mid = (c(W1.data_ptr()), W1.stride(0), c(_b1.data_ptr()), int(D))
        self.tail = (self.N, P.shape[1], T, self.dims, self.Wp, self.bp, c(wo_.data_ptr()), c(bo_.data_ptr()))
        self.fn = getattr(B_.load(), 'cdr_conet_fullsort_users')

    def fresh(self, user_table, W1, b1, weights, biases, wo, bo):
        """The pointers packed at construction are still the model's (a parameter re-allocated since -- ``.to()``, ``.data = ...`` -- makes the
        pack stale; in-place updates keep it valid)."""
        return (self.table_ptr == user_table.data_ptr()
                and self.param_ptrs == tuple(t.data_ptr() for t in [W1, b1, wo, bo] + list(weights) + list(biases)))

    def takes(self, uid):
        U = uid.shape[0]
        return U <= self.MAX_USERS and U * self.N <= self.MAX_PAIRS and uid.dtype == torch.int64 and uid.is_contiguous() and uid.device == self.dev

    def __call__(self, uid):
        """uid: int64 device tensor [U], contiguous."""
        U = uid.shape[0]
        out = torch.empty(U, self.N, device=self.dev, dtype=torch.float32)
        rc = self.fn(B_.stream(), *self.head, B_._c_ptr(uid.data_ptr()), *self.mid, U, *self.tail, B_._c_ptr(out.data_ptr()), self.N)
        if rc:
            B_._check(rc, 'cdr_conet_fullsort_users')
        return out


class ConetFusedLoss(Function):
    """CoNet.calculate_loss (conet.py:183-203) as ONE autograd node on the fused tower kernels: forward = gather + every
    cross unit of both towers + output units + BCE x2 + sum ||H_l||_F in one launch (+ a finishing block); backward = data
    gradients, weight gradients (fixed-order reduction, no float atomics), then the dense embedding gradients the
    reference's caller expects for ``sparse=False`` tables.  ``params`` = for every layer (Ws, bs, Wt, bt, H), then
    (wo_s, bo_s, wo_t, bo_t)."""

    @staticmethod
    def forward(ctx, su, si, tu, ti, user_s, item_s, label_s, user_t, item_t, label_t, n_overlap, overlap_users, dims, row_opt, *params):
        _dev_check(su, si, tu, ti, user_s, item_s, label_s, user_t, item_t, label_t, *params)
        user_s, item_s, user_t, item_t = _ids(user_s), _ids(item_s), _ids(user_t), _ids(item_t)
        label_s = label_s.reshape(-1).contiguous().to(torch.float32)
        label_t = label_t.reshape(-1).contiguous().to(torch.float32)
        n_source = user_s.numel()
        R, D, L = n_source + user_t.numel(), su.shape[1], len(dims) - 1
        dev = su.device
        params = tuple(p.contiguous() for p in params)
        dims_c = (ctypes.c_int * (L + 1))(*[int(d) for d in dims])
        aw, need = ctypes.c_int(0), ctypes.c_size_t(0)
        B_._check(B_.load().cdr_conet_plan(L, dims_c, R, ctypes.byref(aw), ctypes.byref(need)), 'cdr_conet_plan')
        f32 = lambda *shape: torch.empty(*shape, device=dev, dtype=torch.float32)
        x0, acts, prob, maskf, label, out = f32(R, 4 * D), f32(R, aw.value), f32(R), f32(R), f32(R), f32(4 + L)
        ids = torch.empty(2 * R, device=dev, dtype=torch.int64)          # the stacked user ids, then the stacked item ids
        pp = (ctypes.c_void_p * len(params))(*[p.data_ptr() for p in params])
        # a forward that will be differentiated also runs the data backward of every row block, in the same launch (for a unit upstream
        # gradient; ``backward`` applies any other): the activations never leave LDS in between
        train = any(ctx.needs_input_grad) and os.environ.get('CDR_CONET_TWO_LAUNCH', '0') != '1'     # (the switch is for A/B runs and tests)
        gz = gx0 = ws = None
        if train:
            # (a workspace of its own: its output-unit partials must survive until THIS node's backward, whatever runs in between)
            gz, gx0, ws = f32(R, aw.value), f32(R, 4 * D), torch.empty(int(need.value), device=dev, dtype=torch.uint8)
        done = ctypes.c_int(0)
        # a captured, pipelined step (graph_step.GraphedTrainStep) differentiates every loss at once and reads it afterwards: the addition of
        # the forward blocks' loss partials then rides in the backward's weight-gradient launch (cdr_conet_defer_finish; ``out`` -- and the
        # loss tensor -- are complete only behind that launch).  Everyone else gets the loss when the forward's launches retire.
        B_.call('cdr_conet_defer_finish', B_.ctx(dev), 1 if (train and row_opt is not None and getattr(row_opt, 'defer_finish', False)) else 0)
        B_.call('cdr_conet_fwd', B_.ctx(dev), B_.stream(), B_.f32(su), B_.f32(si), B_.f32(tu), B_.f32(ti), D, B_.i64(user_s),
                B_.i64(user_t), B_.i64(item_s), B_.i64(item_t), R, int(n_source), int(n_overlap), 1 if overlap_users else 0, L, dims_c,
                pp, B_.f32(label_s), B_.f32(label_t), B_.f32(x0), B_.f32(acts), B_.f32(prob), B_.f32(maskf), B_.f32(label),
                B_.i64(ids), B_.f32(out), B_.f32(gz) if train else None, B_.f32(gx0) if train else None, B_.raw(ws) if train else None,
                ws.numel() if train else 0, ctypes.byref(done))
        ctx.save_for_backward(ids, label, x0, acts, prob, maskf, out, *params)
        ctx.data_done = (gz, gx0, ws) if done.value else None
        if row_opt is not None:
            # the per-occurrence gradient rows of a unit upstream gradient exist already (the forward launch ran the data backward): a
            # pipelined captured step may launch the row update now (lazyadam.DeferredRowAdam.apply_early)
            row_opt.early = (gx0, (0, D, 2 * D, 3 * D), 4 * D) if done.value else None
        ctx.meta = (int(n_source), tuple(int(d) for d in dims), aw.value, int(need.value), tuple(su.shape), tuple(si.shape))
        ctx.row_opt = row_opt
        ctx.mark_non_differentiable(out)
        ctx.set_materialize_grads(False)      # no zero-filled gradient for the non-differentiable output (a launch per step)
        return out[0], out

    @staticmethod
    def backward(ctx, grad_loss, _g_out):
        ids, label, x0, acts, prob, maskf, out = ctx.saved_tensors[:7]
        params = ctx.saved_tensors[7:]
        n_source, dims, aw, need, ushape, ishape = ctx.meta
        R, L, D = label.numel(), len(dims) - 1, ushape[1]
        user, item = ids[:R], ids[R:]
        dev = x0.device
        done = ctx.data_done is not None
        if done:
            gz, gx0, ws = ctx.data_done      # made by the forward's launch; the workspace holds its output-unit partials
            ctx.data_done = None
        else:
            ws = _conet_workspace(dev, need)
            gz = torch.empty(R, aw, device=dev, dtype=torch.float32)
            gx0 = torch.empty(R, 4 * D, device=dev, dtype=torch.float32)
        grads = tuple(torch.empty_like(p) for p in params)
        dims_c = (ctypes.c_int * (L + 1))(*dims)
        pp = (ctypes.c_void_p * len(params))(*[p.data_ptr() for p in params])
        gp = (ctypes.c_void_p * len(grads))(*[g.data_ptr() for g in grads])
        go = grad_loss.reshape(-1).contiguous().to(torch.float32)
        B_.call('cdr_conet_bwd', B_.ctx(dev), B_.stream(), R, n_source, L, dims_c, pp, B_.f32(label), B_.f32(x0), B_.f32(acts),
                B_.f32(prob), B_.f32(maskf), B_.f32(out), B_.f32(go), B_.f32(gz), B_.f32(gx0), gp, B_.raw(ws), ws.numel(), 1 if done else 0)
        del pp, gp
        if ctx.row_opt is not None:
            # deferred row-wise Adam (lazyadam.DeferredRowAdam): the tables get no dense gradient at all -- the optimizer reads
            # the per-occurrence rows of gx0 through the id sort it made before the forward pass
            ctx.row_opt.pending = (gx0, (0, D, 2 * D, 3 * D), 4 * D)
            return (None,) * 14 + grads
        # dense gradients for the reference's dense optimizer, WITHOUT float atomics: one id sort per list, then every distinct row
        # is written once with its occurrences summed in occurrence order (run-to-run reproducible)
        gsu, gtu = torch.zeros(ushape, device=dev), torch.zeros(ushape, device=dev)
        gsi, gti = torch.zeros(ishape, device=dev), torch.zeros(ishape, device=dev)
        (ku, pu, _), (ki, pi, _) = sort_id_lists([user, item], max(ushape[0], ishape[0]))
        for k, (g, kk, pp_) in enumerate(((gsu, ku, pu), (gsi, ki, pi), (gtu, ku, pu), (gti, ki, pi))):
            B_.call('cdr_scatter_rows_sorted', B_.stream(), B_.f32(g), D, B_.raw(kk), B_.raw(pp_), R, B_._c_ptr(gx0.data_ptr() + 4 * k * D),
                    4 * D)
        return (gsu, gsi, gtu, gti) + (None,) * 10 + grads


# ---------------------------------------------------------------------------------------------------- SSCDR pieces
class GatherMapRows(Function):
    """The four row sets of SSCDR's map phase (sscdr.py:161-168) in ONE launch, and their dense backward in one zero-fill + one scatter
    launch: ``X3`` [3 n, D] = [source_tab[idx] ; other_tab[pos] ; other_tab[neg]] (what the mapping is applied to), ``Tt`` [n, D] =
    target_tab[idx].  ``bump``: optional device int64 advanced in the gather's launch (the sampler's call counter)."""

    @staticmethod
    def forward(ctx, source_tab, target_tab, other_tab, idx, pos, neg, bump):
        _dev_check(source_tab, target_tab, other_tab, idx, pos, neg)
        idx, pos, neg = _ids(idx), _ids(pos), _ids(neg)
        n, D = idx.numel(), source_tab.shape[1]
        dev = source_tab.device
        X3 = torch.empty(3 * n, D, device=dev, dtype=torch.float32)
        Tt = torch.empty(n, D, device=dev, dtype=torch.float32)
        P4, I4 = ctypes.c_void_p * 4, ctypes.c_int64 * 4
        keep = [source_tab, target_tab, other_tab, idx, pos, neg, X3, Tt]
        B_.call('cdr_gather_rows_multi', B_.stream(), 4, P4(source_tab.data_ptr(), other_tab.data_ptr(), other_tab.data_ptr(), target_tab.data_ptr()),
                D, P4(idx.data_ptr(), pos.data_ptr(), neg.data_ptr(), idx.data_ptr()), I4(n, n, n, n),
                P4(X3.data_ptr(), X3.data_ptr() + 4 * n * D, X3.data_ptr() + 8 * n * D, Tt.data_ptr()), None if bump is None else B_.i64(bump))
        del keep
        ctx.save_for_backward(idx, pos, neg)
        ctx.shapes = (tuple(source_tab.shape), tuple(target_tab.shape), tuple(other_tab.shape))
        return X3, Tt

    @staticmethod
    def backward(ctx, gX3, gT):
        idx, pos, neg = ctx.saved_tensors
        ss, ts, os_ = ctx.shapes
        n, D = idx.numel(), ss[1]
        dev = idx.device
        if _ord_fits(D, 2 * n) and ss[1] == ts[1] == os_[1]:
            ns_, nt_, no_ = ss[0] * D, ts[0] * D, os_[0] * D
            flat = torch.zeros(ns_ + nt_ + no_, device=dev, dtype=torch.float32)
            gs, gt, go_ = flat[:ns_].view(ss), flat[ns_:ns_ + nt_].view(ts), flat[ns_ + nt_:].view(os_)
            gX3 = None if gX3 is None else gX3.contiguous()
            gT = None if gT is None else gT.contiguous()
            lists = []
            if gX3 is not None:
                x0 = gX3.data_ptr()
                lists.append((gs.data_ptr(), D, [_ord_seg(idx, X=x0, x_stride=D)]))
                lists.append((go_.data_ptr(), D, [_ord_seg(pos, X=x0 + 4 * n * D, x_stride=D), _ord_seg(neg, X=x0 + 8 * n * D, x_stride=D)]))
            if gT is not None:
                lists.append((gt.data_ptr(), D, [_ord_seg(idx, X=gT.data_ptr(), x_stride=D)]))
            _ordered_bwd(D, lists, keep=(flat, gX3, gT))
            return gs, gt, go_, None, None, None, None
        if deterministic():
            zx, zt = (lambda: torch.zeros(3 * n, D, device=dev)), (lambda: torch.zeros(n, D, device=dev))
            gX3 = zx() if gX3 is None else gX3.contiguous()
            gT = zt() if gT is None else gT.contiguous()
            return (_scatter_rows_deterministic(ss, idx, gX3[:n]), _scatter_rows_deterministic(ts, idx, gT),
                    _scatter_rows_deterministic(os_, torch.cat([pos, neg]), gX3[n:]), None, None, None, None)
        ns, nt, no = ss[0] * D, ts[0] * D, os_[0] * D
        flat = torch.zeros(ns + nt + no, device=dev, dtype=torch.float32)              # the three dense gradients out of ONE fill
        gs, gt, go_ = flat[:ns].view(ss), flat[ns:ns + nt].view(ts), flat[ns + nt:].view(os_)
        P4, I4 = ctypes.c_void_p * 4, ctypes.c_int64 * 4
        gX3 = None if gX3 is None else gX3.contiguous()
        gT = None if gT is None else gT.contiguous()
        x0 = 0 if gX3 is None else gX3.data_ptr()
        nx = n if gX3 is not None else 0
        keep = [flat, gX3, gT, idx, pos, neg]
        B_.call('cdr_scatter_add_rows_multi', B_.stream(), 4, P4(gs.data_ptr(), go_.data_ptr(), go_.data_ptr(), gt.data_ptr()), D,
                P4(idx.data_ptr(), pos.data_ptr(), neg.data_ptr(), idx.data_ptr()), I4(nx, nx, nx, n if gT is not None else 0),
                P4(x0, x0 + 4 * n * D, x0 + 8 * n * D, 0 if gT is None else gT.data_ptr()))
        del keep
        return gs, gt, go_, None, None, None, None


class SSCDRMapLoss(Function):
    """SSCDR.calculate_map_loss's arithmetic after the mapping (sscdr.py:165-172) as ONE node: MSE + lambda x triplet on squared-norm
    normalised rows, its gradients made in the same pass for a unit upstream gradient and rescaled in ``backward`` only if another one
    arrives.  Returns (total [], parts [3] = total, MSE, triplet term)."""

    @staticmethod
    def forward(ctx, mapped3, target_rows, margin, lamda):
        _dev_check(mapped3, target_rows)
        m3, tt = mapped3.contiguous(), target_rows.contiguous()
        n, D = tt.shape
        assert tuple(m3.shape) == (3 * n, D)
        out3 = torch.empty(3, device=m3.device, dtype=torch.float32)
        g3, gt = torch.empty_like(m3), torch.empty_like(tt)
        B_.call('cdr_sscdr_map_loss', B_.ctx(m3.device), B_.stream(), B_.f32(m3), B_.f32(tt), n, D, float(margin), 1e-6, float(lamda),
                B_.f32(out3), B_.f32(g3), B_.f32(gt))
        ctx.save_for_backward(g3, gt)
        ctx.mark_non_differentiable(out3)
        ctx.set_materialize_grads(False)
        return out3[0], out3

    @staticmethod
    def backward(ctx, go, _parts):
        g3, gt = ctx.saved_tensors
        if getattr(ctx, 'consumed', False):          # the stored gradients are rescaled in place: a second pass would scale them twice
            raise RuntimeError('SSCDRMapLoss: backward through this node a second time (retain_graph) is not supported')
        ctx.consumed = True
        go = go.reshape(-1)[:1].contiguous().to(torch.float32)
        B_.call('cdr_scale2_unless_one', B_.stream(), B_.f32(go), B_.f32(g3), g3.numel(), B_.f32(gt), gt.numel())
        return g3, gt, None, None


class SqnormNormalize(Function):
    """SSCDR.embedding_normalize (sscdr.py:120-124): e / max(sum e^2, 1) -- the SQUARED length, quirk kept."""

    @staticmethod
    def forward(ctx, x):
        _dev_check(x)
        x_ = x.contiguous()
        y = torch.empty_like(x_)
        ln = torch.empty(x_.shape[0], device=x.device, dtype=torch.float32)
        B_.call('cdr_sqnorm_normalize_fwd', B_.stream(), B_.f32(x_), x_.shape[0], x_.shape[1], B_.f32(y), B_.f32(ln))
        ctx.save_for_backward(x_, ln)
        return y

    @staticmethod
    def backward(ctx, gy):
        x_, ln = ctx.saved_tensors
        gx = torch.empty_like(x_)
        B_.call('cdr_sqnorm_normalize_bwd', B_.stream(), B_.f32(x_), B_.f32(ln), B_.f32(gy.contiguous()), x_.shape[0],
                x_.shape[1], B_.f32(gx))
        return gx


def sqnorm_normalize(x):
    return SqnormNormalize.apply(x)


class TripletMarginLoss(Function):
    """nn.TripletMarginLoss(margin) with torch's defaults p=2, eps=1e-6, mean (sscdr.py:69)."""

    @staticmethod
    def forward(ctx, a, p, n, margin):
        _dev_check(a, p, n)
        a_, p_, n_ = a.contiguous(), p.contiguous(), n.contiguous()
        rows, D = a_.shape
        out = torch.empty(1, device=a.device, dtype=torch.float32)
        dap = torch.empty(rows, device=a.device, dtype=torch.float32)
        dan = torch.empty(rows, device=a.device, dtype=torch.float32)
        B_.call('cdr_triplet_fwd', B_.ctx(a.device), B_.stream(), B_.f32(a_), B_.f32(p_), B_.f32(n_), rows, D, float(margin),
                1e-6, B_.f32(out), B_.f32(dap), B_.f32(dan))
        ctx.save_for_backward(a_, p_, n_, dap, dan)
        ctx.margin = float(margin)
        return out.reshape(())

    @staticmethod
    def backward(ctx, go):
        a_, p_, n_, dap, dan = ctx.saved_tensors
        ga, gp, gn = torch.empty_like(a_), torch.empty_like(p_), torch.empty_like(n_)
        B_.call('cdr_triplet_bwd', B_.stream(), B_.f32(a_), B_.f32(p_), B_.f32(n_), a_.shape[0], a_.shape[1], ctx.margin, 1e-6,
                B_.f32(dap), B_.f32(dan), B_.f32(go.reshape(-1).contiguous()), B_.f32(ga), B_.f32(gp), B_.f32(gn))
        return ga, gp, gn, None


# ---------------------------------------------------------------------------------------------------- BiTGCF pieces
class CSRGraph:
    """Device CSR of one domain's normalised adjacency + the degree vectors of the transfer layer."""

    def __init__(self, indptr, indices, values, n_rows):
        self.indptr, self.indices, self.values, self.n_rows = indptr, indices, values, n_rows


def _dropout(x, n, p, seed, salt):
    """In-place counter-based dropout; ``seed`` a host int or a device int64 [1] counter."""
    if torch.is_tensor(seed):
        B_.call('cdr_dropout_dev', B_.stream(), B_.f32(x), n, float(p), B_.i64(seed), int(salt), B_.f32(x))
    else:
        B_.call('cdr_dropout', B_.stream(), B_.f32(x), n, float(p), int(seed + salt), B_.f32(x))


class BiTGCFPropagate(Function):
    """BiTGCF.forward (bitgcf.py:174-205) as ONE autograd node: n_layers x [graph layer (CSR SpMM with the elementwise
    math fused) -> bi-directional transfer on the overlapped rows -> L2-normalised copy into the layer stack], both
    domains, then concat / mean.  Backward replays the chain in reverse with the saved per-layer tensors; the adjacency
    is symmetric, so the SpMM backward is the same kernel."""

    @staticmethod
    def forward(ctx, su, si, tu, ti, gs, gt, deg, n_layers, lam_s, lam_t, connect_way, OU, OI, drop_p=0.0, drop_seed=0, rows_hint=None,
                emb_loss=False):
        # drop_seed: a host int, or a device int64 [1] counter (the capturable form: a hipGraph replay would bake a host seed
        # into its launches and repeat one mask on every step)
        # rows_hint = (user ids, item ids, user ids, item ids, ...): the ONLY rows of the returned stacks the caller will read
        # (calculate_loss gathers the batch's users and items, nothing else: bitgcf.py:222-240).  The last layer's output feeds
        # nothing but those rows, so its graph layer, its transfer / normalise and their backward run on the flagged rows only
        # (a tenth of the table at BASELINE C4): same numbers in the flagged rows, same gradients bit for bit (the skipped terms
        # are exact zeros); every other row of the last layer's block reads 0.  Without the hint every row is computed.
        # emb_loss (with rows_hint = (source users, source items, target users, target items)): two more outputs, recbole's EmbLoss of
        # the two batches' EGO rows (bitgcf.py:231-233, as functional.EmbLossRows); their gradient rows are added straight into the
        # table gradients this node returns -- as separate nodes they cost two zero-filled tables and four table-sized adds per step.
        _dev_check(su, si, tu, ti)
        nu, ni, D = su.shape[0], si.shape[0], su.shape[1]
        n = nu + ni
        dev = su.device
        st = B_.stream
        nb = n_layers + 1
        f32 = lambda *shape: torch.empty(*shape, device=dev, dtype=torch.float32)
        flags = None
        if rows_hint is not None and n_layers >= 1:
            lists = [_ids(x) for x in rows_hint]
            m = len(lists)
            assert m % 2 == 0, 'rows_hint = (users, items) pairs'
            need = ctypes.c_size_t(0)
            B_._check(B_.load().cdr_row_flags_layout(n, ctypes.byref(need)), 'cdr_row_flags_layout')
            flags = torch.empty(int(need.value), device=dev, dtype=torch.uint8)      # byte flags, then the bit map and the row list
            B_._alive.extend(lists)
            B_.call('cdr_row_flags', st(), m, (ctypes.c_void_p * m)(*[x.data_ptr() for x in lists]), (ctypes.c_int64 * m)(*[x.numel() for x in lists]),
                    (ctypes.c_int64 * m)(*[0 if k % 2 == 0 else nu for k in range(m)]), n, B_.raw(flags), flags.numel())
        S, T, catS, catT = f32(n, D), f32(n, D), f32(n, nb * D), f32(n, nb * D)
        B_.call('cdr_bitgcf_stack', st(), B_.f32(su.contiguous()), B_.f32(si.contiguous()), B_.f32(tu.contiguous()), B_.f32(ti.contiguous()),
                nu, ni, D, B_.f32(S), B_.f32(T), B_.f32(catS), B_.f32(catT), nb * D)
        saved = []
        for l in range(n_layers):
            sideS, newS, sideT, newT = f32(n, D), f32(n, D), f32(n, D), f32(n, D)
            fl = B_.raw(flags) if (flags is not None and l == n_layers - 1) else None
            B_.call('cdr_graph_layer_fwd', st(), B_.i64(gs.indptr), B_.i64(gs.indices), B_.f32(gs.values), n, B_.f32(S), D,
                    B_.f32(sideS), B_.f32(newS), fl)
            B_.call('cdr_graph_layer_fwd', st(), B_.i64(gt.indptr), B_.i64(gt.indices), B_.f32(gt.values), n, B_.f32(T), D,
                    B_.f32(sideT), B_.f32(newT), fl)
            # [dropout ->] transfer -> L2-normalised copy into the layer stack: one launch for users and items of both domains
            S2, T2, nS, nT = f32(n, D), f32(n, D), f32(n), f32(n)
            dev_seed = torch.is_tensor(drop_seed)
            B_.call('cdr_bitgcf_mix_fwd', st(), B_.f32(newS), B_.f32(newT), B_.f32(deg['su']), B_.f32(deg['tu']), B_.f32(deg['si']),
                    B_.f32(deg['ti']), nu, ni, D, OU, OI, lam_s, lam_t, float(drop_p), 0 if dev_seed else int(drop_seed),
                    B_.i64(drop_seed) if dev_seed else None, 2 * l, 2 * l + 1, B_.f32(S2), B_.f32(T2),
                    B_._c_ptr(catS.data_ptr() + 4 * (l + 1) * D), B_._c_ptr(catT.data_ptr() + 4 * (l + 1) * D), nb * D, B_.f32(nS), B_.f32(nT), fl)
            saved += [S, T, sideS, sideT, S2, T2, nS, nT]
            S, T = S2, T2
        if connect_way == 'concat':
            outS, outT = catS, catT
        else:
            outS, outT = f32(n, D), f32(n, D)
            B_.call('cdr_colblock_mean_fwd', st(), B_.f32(catS), n, D, nb, B_.f32(outS))
            B_.call('cdr_colblock_mean_fwd', st(), B_.f32(catT), n, D, nb, B_.f32(outT))
        embS = embT = None
        ctx.emb = None
        if emb_loss:
            assert rows_hint is not None and len(rows_hint) == 4, 'emb_loss needs rows_hint = (source users, source items, target users, target items)'
            ids4 = [_ids(x) for x in rows_hint]
            su_, si_, tu_, ti_ = su.contiguous(), si.contiguous(), tu.contiguous(), ti.contiguous()
            o3s, o3t = f32(3), f32(3)
            B_.call('cdr_embloss_fwd', B_.ctx(dev), st(), B_.f32(su_), B_.f32(si_), D, B_.i64(ids4[0]), B_.i64(ids4[1]), ids4[0].numel(), B_.f32(o3s))
            B_.call('cdr_embloss_fwd', B_.ctx(dev), st(), B_.f32(tu_), B_.f32(ti_), D, B_.i64(ids4[2]), B_.i64(ids4[3]), ids4[2].numel(), B_.f32(o3t))
            ctx.emb = (su_, si_, tu_, ti_, ids4, o3s, o3t)
            embS, embT = o3s[:1], o3t[:1]
        ctx.save_for_backward(*saved)
        ctx.meta = (gs, gt, deg, n_layers, lam_s, lam_t, connect_way, OU, OI, nu, ni, D, float(drop_p), drop_seed)
        ctx.flags = flags
        ctx.set_materialize_grads(False)
        return outS, outT, embS, embT

    @staticmethod
    def backward(ctx, gOutS, gOutT, gEmbS=None, gEmbT=None):
        gs, gt, deg, n_layers, lam_s, lam_t, connect_way, OU, OI, nu, ni, D, drop_p, drop_seed = ctx.meta
        saved = ctx.saved_tensors
        n, nb = nu + ni, n_layers + 1
        dev = saved[0].device if len(saved) else (gOutS if gOutS is not None else gOutT).device
        if gOutS is None or gOutT is None:       # (an unused stack: its gradient is zero)
            z = lambda: torch.zeros(n, nb * D if connect_way == 'concat' else D, device=dev, dtype=torch.float32)
            gOutS = z() if gOutS is None else gOutS
            gOutT = z() if gOutT is None else gOutT
        st = B_.stream
        f32 = lambda *shape: torch.empty(*shape, device=dev, dtype=torch.float32)
        if connect_way == 'concat':
            gcatS, gcatT = gOutS.contiguous(), gOutT.contiguous()
        else:
            gcatS, gcatT = f32(n, nb * D), f32(n, nb * D)
            B_.call('cdr_colblock_mean_bwd', st(), B_.f32(gOutS.contiguous()), n, D, nb, B_.f32(gcatS))
            B_.call('cdr_colblock_mean_bwd', st(), B_.f32(gOutT.contiguous()), n, D, nb, B_.f32(gcatT))
        gS = gT = None                        # gradient w.r.t. the un-normalised layer output that continues downward
        tmp = f32(n, D)
        for l in reversed(range(n_layers)):
            S_in, T_in, sideS, sideT, S2, T2, nS, nT = saved[8 * l:8 * l + 8]
            fl = B_.raw(ctx.flags) if (ctx.flags is not None and l == n_layers - 1) else None
            # backward of the same chain in one launch: normalise -> (+ gradient from the layer above) -> transfer [-> dropout mask]
            gnS, gnT = f32(n, D), f32(n, D)
            dev_seed = torch.is_tensor(drop_seed)
            B_.call('cdr_bitgcf_mix_bwd', st(), B_.f32(S2), B_.f32(T2), B_.f32(nS), B_.f32(nT),
                    B_._c_ptr(gcatS.data_ptr() + 4 * (l + 1) * D), B_._c_ptr(gcatT.data_ptr() + 4 * (l + 1) * D), nb * D,
                    B_.f32(gS), B_.f32(gT), B_.f32(deg['su']), B_.f32(deg['tu']), B_.f32(deg['si']), B_.f32(deg['ti']), nu, ni, D, OU, OI,
                    lam_s, lam_t, float(drop_p), 0 if dev_seed else int(drop_seed), B_.i64(drop_seed) if dev_seed else None,
                    2 * l, 2 * l + 1, B_.f32(gnS), B_.f32(gnT), fl)
            gS_in, gT_in = f32(n, D), f32(n, D)
            B_.call('cdr_graph_layer_bwd', st(), B_.i64(gs.indptr), B_.i64(gs.indices), B_.f32(gs.values), n, B_.f32(S_in),
                    B_.f32(sideS), B_.f32(gnS), D, B_.f32(tmp), B_.f32(gS_in), fl)
            B_.call('cdr_graph_layer_bwd', st(), B_.i64(gt.indptr), B_.i64(gt.indices), B_.f32(gt.values), n, B_.f32(T_in),
                    B_.f32(sideT), B_.f32(gnT), D, B_.f32(tmp), B_.f32(gT_in), fl)
            gS, gT = gS_in, gT_in
        if gS is None:
            gS, gT = torch.zeros(n, D, device=dev), torch.zeros(n, D, device=dev)
        # layer-0 block of the stack is the ego embedding itself
        B_.call('cdr_bitgcf_unstack_bwd', st(), B_.f32(gcatS), B_.f32(gcatT), nb * D, n, D, B_.f32(gS), B_.f32(gT))
        if ctx.emb is not None:
            su_, si_, tu_, ti_, ids4, o3s, o3t = ctx.emb
            for g, (U, I, u, i, o3, go) in ((gS, (su_, si_, ids4[0], ids4[1], o3s, gEmbS)), (gT, (tu_, ti_, ids4[2], ids4[3], o3t, gEmbT))):
                if go is not None:
                    B_.call('cdr_embloss_bwd_dense', st(), B_.f32(U), B_.f32(I), D, B_.i64(u), B_.i64(i), u.numel(), B_.f32(o3),
                            B_.f32(go.reshape(-1).contiguous()), B_.f32(g), B_._c_ptr(g.data_ptr() + 4 * nu * D))
        return (gS[:nu], gS[nu:], gT[:nu], gT[nu:]) + (None,) * 13


class _InnerCtx:
    """What a Function's forward / backward need of ``ctx`` when another node runs them as a part of itself (BiTGCFLoss)."""
    needs_input_grad = (True,)

    def save_for_backward(self, *tensors):
        self.saved_tensors = tensors

    def set_materialize_grads(self, _flag):
        pass


class BiTGCFLoss(Function):
    """BiTGCF.calculate_loss (bitgcf.py:207-240) as ONE autograd node: the propagation (BiTGCFPropagate's forward, last layer on the
    batch's rows), then both domains' losses -- BCE on the batch's rows of the propagated stacks + reg_weight x EmbLoss of the batch's EGO
    rows -- in ONE launch (cdr_point_fwd_pair_ex: the EmbLoss rows ride in the same lane groups as the stack rows; the finishing block
    writes bce + reg_weight (||U_b|| + ||I_b||) / B).  Backward: one scatter launch into the stack gradients (zero-filled on the side by
    the loss launch), the propagation's backward, one launch for both domains' EmbLoss gradient rows added into the table gradients.
    Against separate nodes (round 4): no EmbLoss partial / finish launches (4), no scalar adds and their backward scalings (4), one EmbLoss
    backward launch instead of two, no fill launch: 37 -> 27 launches per step at BASELINE C4.  Returns (loss_source [1], loss_target [1])."""

    @staticmethod
    def forward(ctx, su, si, tu, ti, gs, gt, deg, n_layers, lam_s, lam_t, connect_way, OU, OI, drop_p, drop_seed, reg_weight,
                us, is_, ls, ut, it, lt):
        dev, D = su.device, su.shape[1]
        nu = su.shape[0]
        inner = _InnerCtx()
        S, T, _, _ = BiTGCFPropagate.forward(inner, su, si, tu, ti, gs, gt, deg, n_layers, lam_s, lam_t, connect_way, OU, OI, drop_p, drop_seed,
                                             (us, is_, ut, it), False)
        W = S.shape[1]
        assert W % 4 == 0 and D % 4 == 0 and S.is_contiguous() and T.is_contiguous()
        ids = [_ids(us), _ids(is_), _ids(ut), _ids(it)]
        labels = [ls.reshape(-1).contiguous().to(torch.float32), lt.reshape(-1).contiguous().to(torch.float32)]
        ego = [su.contiguous(), si.contiguous(), tu.contiguous(), ti.contiguous()]
        out8 = torch.empty(2, 4, device=dev, dtype=torch.float32)
        gc = [torch.empty(ids[0].numel(), device=dev, dtype=torch.float32), torch.empty(ids[2].numel(), device=dev, dtype=torch.float32)]
        gstack = None
        if any(ctx.needs_input_grad[:4]):
            gstack = torch.empty(2, S.shape[0], W, device=dev, dtype=torch.float32)          # zero-filled by the loss launch, on the side
            B_.call('cdr_ctx_scrub_next', B_.ctx(dev), B_.raw(gstack), 4 * gstack.numel())
        P2, I2, F2 = ctypes.c_void_p * 2, ctypes.c_int64 * 2, ctypes.c_float * 2
        io = 4 * nu * W                                               # byte offset of the item rows inside a stack
        B_._alive.extend(ids + labels + ego)
        B_.call('cdr_point_fwd_pair_ex', B_.ctx(dev), B_.stream(), B_.CDR_LOSS_BCE, P2(S.data_ptr(), T.data_ptr()), P2(S.data_ptr() + io, T.data_ptr() + io),
                P2(ego[0].data_ptr(), ego[2].data_ptr()), P2(ego[1].data_ptr(), ego[3].data_ptr()), W, D,
                P2(ids[0].data_ptr(), ids[2].data_ptr()), P2(ids[1].data_ptr(), ids[3].data_ptr()), P2(labels[0].data_ptr(), labels[1].data_ptr()),
                I2(ids[0].numel(), ids[2].numel()), F2(float(reg_weight), float(reg_weight)), P2(out8.data_ptr(), out8.data_ptr() + 16),
                P2(gc[0].data_ptr(), gc[1].data_ptr()), None, None, None)
        ctx.inner, ctx.keep = inner, (S, T, ids, gc, out8, ego, gstack)
        ctx.nu, ctx.reg_weight = nu, float(reg_weight)
        ctx.set_materialize_grads(False)
        return out8[0, :1], out8[1, :1]

    @staticmethod
    def backward(ctx, g_s, g_t):
        S, T, ids, gc, out8, ego, gstack = ctx.keep
        dev, W, D, nu = S.device, S.shape[1], ego[0].shape[1], ctx.nu
        zero = None
        gos = []
        for g in (g_s, g_t):
            if g is None:
                zero = torch.zeros(1, device=dev, dtype=torch.float32) if zero is None else zero
                g = zero
            gos.append(g.reshape(-1)[:1].contiguous().to(torch.float32))
        if gstack is None:
            gstack = torch.zeros(2, S.shape[0], W, device=dev, dtype=torch.float32)
        ctx.keep = None
        gS, gT = gstack[0], gstack[1]
        P2, I2, F2 = ctypes.c_void_p * 2, ctypes.c_int64 * 2, ctypes.c_float * 2
        io = 4 * nu * W
        nB = I2(ids[0].numel(), ids[2].numel())
        ordered = _ord_fits(W, ids[0].numel(), ids[1].numel(), ids[2].numel(), ids[3].numel())
        # stack gradients only (reg_weight 0 here: the EmbLoss rows belong to the EGO tables and are added at the end, below)
        if ordered:
            lists = []
            for X_, gX_, u, i, c, gop in ((S, gS, ids[0], ids[1], gc[0], gos[0]), (T, gT, ids[2], ids[3], gc[1], gos[1])):
                kw = dict(coef=c.data_ptr(), go=gop.data_ptr(), x_stride=W)
                lists.append((gX_.data_ptr(), W, [_ord_seg(u, X=X_.data_ptr() + io, xid=i.data_ptr(), **kw)]))       # user rows < nu
                lists.append((gX_.data_ptr() + io, W, [_ord_seg(i, X=X_.data_ptr(), xid=u.data_ptr(), **kw)]))       # item rows from nu on
            _ordered_bwd(W, lists)
        else:
            B_.call('cdr_point_bwd_dense_pair', B_.ctx(dev), B_.stream(), P2(S.data_ptr(), T.data_ptr()), P2(S.data_ptr() + io, T.data_ptr() + io), None, None, W,
                P2(ids[0].data_ptr(), ids[2].data_ptr()), P2(ids[1].data_ptr(), ids[3].data_ptr()), nB,
                P2(gc[0].data_ptr(), gc[1].data_ptr()), P2(out8.data_ptr(), out8.data_ptr() + 16), F2(0.0, 0.0),
                    P2(gos[0].data_ptr(), gos[1].data_ptr()), None, P2(gS.data_ptr(), gT.data_ptr()), P2(gS.data_ptr() + io, gT.data_ptr() + io), None, None)
        grads = BiTGCFPropagate.backward(ctx.inner, gS, gT, None, None)
        gsu, gsi, gtu, gti = grads[:4]                              # views of two [n, D] buffers: users, then items
        if ctx.reg_weight != 0.0 and ordered and D <= 256 and all(g_.is_contiguous() for g_ in (gsu, gsi, gtu, gti)):
            lists = []
            for d, (gu_, gi_) in enumerate(((gsu, gsi), (gtu, gti))):
                kw = dict(go=gos[d].data_ptr(), r_stride=D, reg=ctx.reg_weight, B=ids[2 * d].numel())
                lists.append((gu_.data_ptr(), D, [_ord_seg(ids[2 * d], R=ego[2 * d].data_ptr(), norm=out8.data_ptr() + 16 * d + 8, **kw)]))
                lists.append((gi_.data_ptr(), D, [_ord_seg(ids[2 * d + 1], R=ego[2 * d + 1].data_ptr(), norm=out8.data_ptr() + 16 * d + 12, **kw)]))
            _ordered_bwd(D, lists, accumulate=True)
        elif ctx.reg_weight != 0.0:
            B_.call('cdr_embloss_bwd_dense_pair', B_.stream(), P2(ego[0].data_ptr(), ego[2].data_ptr()), P2(ego[1].data_ptr(), ego[3].data_ptr()), D,
                    P2(ids[0].data_ptr(), ids[2].data_ptr()), P2(ids[1].data_ptr(), ids[3].data_ptr()), nB,
                    P2(out8.data_ptr() + 8, out8.data_ptr() + 24), P2(gos[0].data_ptr(), gos[1].data_ptr()), F2(ctx.reg_weight, ctx.reg_weight),
                    P2(gsu.data_ptr(), gtu.data_ptr()), P2(gsi.data_ptr(), gti.data_ptr()))
        del gos
        ctx.inner = None
        return (gsu, gsi, gtu, gti) + (None,) * 18


class EmbLossRows(Function):
    """recbole EmbLoss of gathered EGO rows: (||U[uid]||_F + ||I[iid]||_F) / B  (bitgcf.py:231-233)."""

    @staticmethod
    def forward(ctx, U, I, uid, iid):
        _dev_check(U, I, uid, iid)
        uid, iid = _ids(uid), _ids(iid)
        out3 = torch.empty(3, device=U.device, dtype=torch.float32)
        B_.call('cdr_embloss_fwd', B_.ctx(U.device), B_.stream(), B_.f32(U), B_.f32(I), U.shape[1], B_.i64(uid), B_.i64(iid),
                uid.numel(), B_.f32(out3))
        ctx.save_for_backward(U, I, uid, iid, out3)
        return out3[:1]

    @staticmethod
    def backward(ctx, go):
        U, I, uid, iid, out3 = ctx.saved_tensors
        if _ord_fits(U.shape[1], uid.numel(), iid.numel()) and U.is_contiguous() and I.is_contiguous() and U.shape[1] == I.shape[1]:
            n, D = uid.numel(), U.shape[1]
            gU, gI = _zeros_like2(U, I)
            g1 = go.reshape(-1).contiguous().to(torch.float32)
            kw = dict(go=g1.data_ptr(), r_stride=D, reg=1.0, B=n)
            _ordered_bwd(D, [(gU.data_ptr(), D, [_ord_seg(uid, R=U.data_ptr(), norm=out3.data_ptr() + 4, **kw)]),
                             (gI.data_ptr(), D, [_ord_seg(iid, R=I.data_ptr(), norm=out3.data_ptr() + 8, **kw)])], keep=(g1, gU, gI))
            return gU, gI, None, None
        if deterministic():
            # the same kernel on the gathered rows with the occurrence index as the id (one add per address), then in-order segment sums
            n, D = uid.numel(), U.shape[1]
            ar = _arange(U.device, n)
            Ur, Ir = _gather_plain(U, uid), _gather_plain(I, iid)
            flat = torch.zeros(2 * n * D, device=U.device, dtype=torch.float32)
            dU, dI = flat[:n * D].view(n, D), flat[n * D:].view(n, D)
            B_.call('cdr_embloss_bwd_dense', B_.stream(), B_.f32(Ur), B_.f32(Ir), D, B_.i64(ar), B_.i64(ar), n,
                    B_.f32(out3), B_.f32(go.reshape(-1).contiguous()), B_.f32(dU), B_.f32(dI))
            return _scatter_rows_deterministic(U.shape, uid, dU), _scatter_rows_deterministic(I.shape, iid, dI), None, None
        gU, gI = _zeros_like2(U, I)
        B_.call('cdr_embloss_bwd_dense', B_.stream(), B_.f32(U), B_.f32(I), U.shape[1], B_.i64(uid), B_.i64(iid), uid.numel(),
                B_.f32(out3), B_.f32(go.reshape(-1).contiguous()), B_.f32(gU), B_.f32(gI))
        return gU, gI, None, None


def bcast_add_act(P, q, act):
    out = torch.empty_like(P)
    B_.call('cdr_bcast_add_act', B_.stream(), B_.f32(P), B_.f32(q.contiguous()), P.shape[0], P.shape[1], int(act), B_.f32(out))
    return out


def adam_dense_(param, grad, exp_avg, exp_avg_sq, step, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
    B_.call('cdr_adam_dense', B_.stream(), B_.f32(param), B_.f32(grad.contiguous()), B_.f32(exp_avg), B_.f32(exp_avg_sq),
            param.numel(), float(lr), float(betas[0]), float(betas[1]), float(eps), float(weight_decay), int(step))


# ---------------------------------------------------------------------------------------------- the five remaining models
class GatherMaxConcat(Function):
    """cat([maximum(Us[u], Ut[u]), maximum(Is[i], It[i])], -1) in one [n, 2D] buffer (dtcdr.py:113-123) with torch.maximum's
    backward (ties split evenly) scatter-added into the four dense table gradients."""

    @staticmethod
    def forward(ctx, us, ut, i_s, it, uid, iid):
        _dev_check(us, ut, i_s, it, uid, iid)
        uid, iid = _ids(uid), _ids(iid)
        n, D = uid.numel(), us.shape[1]
        out = torch.empty(n, 2 * D, device=us.device, dtype=torch.float32)
        B_.call('cdr_gather_max2', B_.stream(), B_.f32(us), B_.f32(ut), D, B_.i64(uid), n, B_.f32(out), 2 * D)
        B_.call('cdr_gather_max2', B_.stream(), B_.f32(i_s), B_.f32(it), D, B_.i64(iid), n, B_._c_ptr(out.data_ptr() + 4 * D), 2 * D)
        ctx.save_for_backward(us, ut, i_s, it, uid, iid)
        return out

    @staticmethod
    def backward(ctx, g):
        us, ut, i_s, it, uid, iid = ctx.saved_tensors
        g = g.contiguous()
        D, n = us.shape[1], uid.numel()
        need = ctx.needs_input_grad
        grads = [torch.zeros_like(w) if need[k] else None for k, w in enumerate((us, ut, i_s, it))]
        if need[0] or need[1]:
            B_.call('cdr_gather_max2_bwd', B_.stream(), B_.f32(us), B_.f32(ut), D, B_.i64(uid), n, B_.f32(g), 2 * D,
                    B_.f32(grads[0]), B_.f32(grads[1]))
        if need[2] or need[3]:
            B_.call('cdr_gather_max2_bwd', B_.stream(), B_.f32(i_s), B_.f32(it), D, B_.i64(iid), n,
                    B_._c_ptr(g.data_ptr() + 4 * D), 2 * D, B_.f32(grads[2]), B_.f32(grads[3]))
        return grads[0], grads[1], grads[2], grads[3], None, None


class ApfProduct(Function):
    """[s (.) t ; o (.) t] -> [2B, D]: both candidates of DeepAPF's attention MLP in one operand (deepapf.py:77-78)."""

    @staticmethod
    def forward(ctx, s, o, t):
        s, o, t = s.contiguous(), o.contiguous(), t.contiguous()
        B, D = s.shape
        X = torch.empty(2 * B, D, device=s.device, dtype=torch.float32)
        B_.call('cdr_apf_prod', B_.stream(), B_.f32(s), B_.f32(o), B_.f32(t), B, D, B_.f32(X))
        ctx.save_for_backward(s, o, t)
        return X

    @staticmethod
    def backward(ctx, gX):
        s, o, t = ctx.saved_tensors
        B, D = s.shape
        gs, go, gt = torch.empty_like(s), torch.empty_like(o), torch.empty_like(t)
        B_.call('cdr_apf_prod_bwd', B_.stream(), B_.f32(s), B_.f32(o), B_.f32(t), B_.f32(gX.contiguous()), B, D,
                B_.f32(gs), B_.f32(go), B_.f32(gt))
        return gs, go, gt


class ApfCombine(Function):
    """masked 2-way softmax over (share, only) scores, attention-merged embedding, predict layer, sigmoid (deepapf.py:80-88)."""

    @staticmethod
    def forward(ctx, a, s, o, t, wp, ids, n_overlap):
        a, s, o, t, wp = a.reshape(-1).contiguous(), s.contiguous(), o.contiguous(), t.contiguous(), wp.reshape(-1).contiguous()
        ids = _ids(ids)
        B, D = s.shape
        p = torch.empty(B, device=s.device, dtype=torch.float32)
        al = torch.empty(B, device=s.device, dtype=torch.float32)
        B_.call('cdr_apf_combine', B_.stream(), B_.f32(a), B_.f32(s), B_.f32(o), B_.f32(t), B_.f32(wp), B_.i64(ids),
                int(n_overlap), B, D, B_.f32(p), B_.f32(al))
        ctx.save_for_backward(s, o, t, wp, p, al)
        ctx.ashape, ctx.wshape = None, None
        return p

    @staticmethod
    def backward(ctx, gp):
        s, o, t, wp, p, al = ctx.saved_tensors
        B, D = s.shape
        dev = s.device
        ga = torch.empty(2 * B, 1, device=dev, dtype=torch.float32)
        gs, go, gt, rows = (torch.empty(B, D, device=dev, dtype=torch.float32) for _ in range(4))
        B_.call('cdr_apf_combine_bwd', B_.stream(), B_.f32(s), B_.f32(o), B_.f32(t), B_.f32(wp), B_.f32(p), B_.f32(al),
                B_.f32(gp.contiguous()), B, D, B_.f32(ga), B_.f32(gs), B_.f32(go), B_.f32(gt), B_.f32(rows))
        return ga, gs, go, gt, _colsum(rows).view(1, D), None, None


class MaxMinNormalize(Function):
    """(x - mean) / (max - mean) per row, mean = (max + min) / 2 (dcdcsr.py:167-172).  Returns (y, stats [n, 2] = mean, max)."""

    @staticmethod
    def forward(ctx, x):
        x = x.contiguous()
        n, D = x.shape
        y = torch.empty_like(x)
        stats = torch.empty(n, 2, device=x.device, dtype=torch.float32)
        B_.call('cdr_maxmin_norm', B_.stream(), B_.f32(x), n, D, B_.f32(y), B_.f32(stats))
        ctx.save_for_backward(x)
        ctx.mark_non_differentiable(stats)
        ctx.set_materialize_grads(False)      # no zero-filled gradient for the non-differentiable output (a launch per step)
        return y, stats

    @staticmethod
    def backward(ctx, gy, _gs):
        (x,) = ctx.saved_tensors
        gx = torch.empty_like(x)
        B_.call('cdr_maxmin_norm_bwd', B_.stream(), B_.f32(x), B_.f32(gy.contiguous()), x.shape[0], x.shape[1], B_.f32(gx))
        return gx


class NatrAttention(Function):
    """NATR's unit-level attention over transferred history rows + domain-level gate + score (natr.py:117-136 / :139-156)."""

    @staticmethod
    def forward(ctx, He, pu, qi, mask, wu, bu, wd, bd):
        He, pu, qi, mask = He.contiguous(), pu.contiguous(), qi.contiguous(), mask.contiguous()
        wu_, wd_ = wu.reshape(-1).contiguous(), wd.reshape(-1).contiguous()
        Bn, L, D = He.shape
        dev = He.device
        att = torch.empty(Bn, L, device=dev, dtype=torch.float32)
        su = torch.empty(Bn, D, device=dev, dtype=torch.float32)
        beta = torch.empty(Bn, device=dev, dtype=torch.float32)
        p = torch.empty(Bn, device=dev, dtype=torch.float32)
        B_.call('cdr_natr_att_fwd', B_.stream(), B_.f32(He), B_.f32(pu), B_.f32(qi), B_.f32(mask), B_.f32(wu_), B_.f32(bu),
                B_.f32(wd_), B_.f32(bd), Bn, L, D, B_.f32(att), B_.f32(su), B_.f32(beta), B_.f32(p))
        ctx.save_for_backward(He, pu, qi, mask, wu_, bu, wd_, bd, att, su, beta, p)
        return p

    @staticmethod
    def backward(ctx, gp):
        He, pu, qi, mask, wu_, bu, wd_, bd, att, su, beta, p = ctx.saved_tensors
        Bn, L, D = He.shape
        dev = He.device
        gHe = torch.empty_like(He)
        gpu, gqi, rwu, rwd = (torch.empty(Bn, D, device=dev, dtype=torch.float32) for _ in range(4))
        rb = torch.empty(Bn, 2, device=dev, dtype=torch.float32)
        B_.call('cdr_natr_att_bwd', B_.stream(), B_.f32(He), B_.f32(pu), B_.f32(qi), B_.f32(mask), B_.f32(wu_), B_.f32(bu),
                B_.f32(wd_), B_.f32(bd), Bn, L, D, B_.f32(att), B_.f32(su), B_.f32(beta), B_.f32(p), B_.f32(gp.contiguous()),
                B_.f32(gHe), B_.f32(gpu), B_.f32(gqi), B_.f32(rwu), B_.f32(rwd), B_.f32(rb))
        gb = _colsum(rb)
        return gHe, gpu, gqi, None, _colsum(rwu).view(1, D), gb[:1], _colsum(rwd).view(1, D), gb[1:]
