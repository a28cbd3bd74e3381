"""Global id remap of a cross-domain dataset (recbole_cdr/data/dataset.py:344-445) and the eval-side "revoke" map
(recbole_cdr/data/dataloader.py:240-247), on the native library (csrc/cdr_remap.cpp).  Bit-exact targets."""
import ctypes
from dataclasses import dataclass

import numpy as np

from .. import binding as B_


@dataclass
class RemapResult:
    source_ids: np.ndarray        # remapped id of every source occurrence (-1 for NaN tokens)
    target_ids: np.ndarray
    num_overlap: int              # PAD included (dataset.py:384)
    num_source_only: int
    num_target_only: int
    num_total: int


def _pack(tokens):
    """list of str (None = NaN) -> (utf-8 bytes, int64 offsets [n+1], uint8 isnan [n])."""
    enc = [b'' if t is None else str(t).encode('utf-8') for t in tokens]
    off = np.zeros(len(enc) + 1, dtype=np.int64)
    np.cumsum([len(e) for e in enc], out=off[1:])
    isnan = np.array([t is None for t in tokens], dtype=np.uint8)
    return b''.join(enc), off, isnan


def overlap_remap_packed(source, target, device):
    """The remap of one field ON THE DEVICE (csrc/cdr_remap_dev.hip: every occurrence of both domains radix-sorted together in byte
    order; dataset.py:344-445 + :109-123).  ``source`` / ``target``: (bytes uint8 [nbytes], offsets int64 [n + 1], isnan uint8 [n] or
    None) as torch tensors or numpy arrays -- tensors already on ``device`` are used in place (a loader that tokenises on the device
    pays no copy).  Returns (source_ids, target_ids) int64 DEVICE tensors (-1 for NaN tokens), counts4 int64 device tensor
    {num_overlap (PAD counted), num_source_only, num_target_only, num_total} and the number of radix passes."""
    import torch
    dev = torch.device(device)

    def put(x, dt):
        if x is None:
            return None
        if not torch.is_tensor(x):
            x = np.ascontiguousarray(x)
            x = torch.from_numpy(x if x.flags.writeable else x.copy())      # (np.frombuffer views of bytes objects are read-only: torch warns)
        t = x
        return t.to(device=dev, dtype=dt).contiguous()

    (sb, so, sn), (tb, to, tn) = [(put(b, torch.uint8), put(o, torch.int64), put(m, torch.uint8)) for b, o, m in (source, target)]
    ns, nt = so.numel() - 1, to.numel() - 1
    sid = torch.empty(max(ns, 0), device=dev, dtype=torch.int64)
    tid = torch.empty(max(nt, 0), device=dev, dtype=torch.int64)
    counts = torch.zeros(4, device=dev, dtype=torch.int64)
    need = ctypes.c_size_t(0)
    B_._check(B_.load().cdr_overlap_remap_dev_workspace_bytes(ns, nt, ctypes.byref(need)), 'cdr_overlap_remap_dev_workspace_bytes')
    ws = torch.empty(int(need.value), device=dev, dtype=torch.uint8)
    passes = ctypes.c_int64(0)
    p = lambda t: None if t is None or t.numel() == 0 else B_.raw(t)
    with torch.cuda.device(dev):
        B_.call('cdr_overlap_remap_dev', B_.stream(), p(sb), B_.raw(so), p(sn), ns, p(tb), B_.raw(to), p(tn), nt, p(sid), p(tid),
                B_.raw(counts), B_.raw(ws), ws.numel(), ctypes.byref(passes))
    return sid, tid, counts, int(passes.value)


def overlap_remap(source_tokens, target_tokens, device=None):
    """One field (users or items) of both domains -> RemapResult.  To reproduce the reference when ``user_feat`` /
    ``item_feat`` exist, append their id columns to the interaction tokens before the call (dataset.py:360-366).
    ``device`` (a ROCm device): the radix-sort form on the GPU (``overlap_remap_packed``) instead of the single-threaded host
    form -- same ids, bit for bit."""
    sb, so, sn = _pack(source_tokens)
    tb, to, tn = _pack(target_tokens)
    if device is not None:
        as_u8 = lambda b: np.frombuffer(b, dtype=np.uint8) if len(b) else np.zeros(0, np.uint8)
        sid, tid, counts, _ = overlap_remap_packed((as_u8(sb), so, sn), (as_u8(tb), to, tn), device)
        c = counts.cpu().numpy()
        return RemapResult(sid.cpu().numpy(), tid.cpu().numpy(), int(c[0]), int(c[1]), int(c[2]), int(c[3]))
    sid = np.empty(len(source_tokens), dtype=np.int64)
    tid = np.empty(len(target_tokens), dtype=np.int64)
    counts = np.zeros(4, dtype=np.int64)
    vp = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    B_.call('cdr_overlap_remap', sb, vp(so), vp(sn), len(source_tokens), tb, vp(to), vp(tn), len(target_tokens),
            vp(sid), vp(tid), vp(counts))
    return RemapResult(sid, tid, int(counts[0]), int(counts[1]), int(counts[2]), int(counts[3]))


def revoke_map(item_ids, overlap_item_num, target_only_item_num):
    """Device tensor of source-domain item ids -> positions in the concatenated source score row
    (``iid if iid < OI else iid - num_target_only_item``)."""
    import torch
    ids = item_ids.contiguous().to(torch.int64)
    out = torch.empty_like(ids)
    if ids.numel():
        B_.call('cdr_revoke_map', B_.stream(), B_.i64(ids), ids.numel(), int(overlap_item_num), int(target_only_item_num),
                B_.i64(out))
    return out
