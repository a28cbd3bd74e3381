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


def overlap_remap(source_tokens, target_tokens):
    """One field (users or items) of both domains -> RemapResult.  To reproduce the reference when ``user_feat`` /
    ``item_feat`` exist, append their id columns to the interaction tokens before the call (dataset.py:360-366)."""
    sb, so, sn = _pack(source_tokens)
    tb, to, tn = _pack(target_tokens)
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
