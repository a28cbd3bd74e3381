"""Full-sort at C5 size, U = 1,024 (MB_U): plain scoring (writes [U, N]) against the fused mask + top-10 (no score matrix), HIP-event timed.
  python tools/mb_fullsort_topk.py [D]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import recbole_cdr_amd  # noqa: F401,E402
from recbole_cdr_amd import functional as F_  # noqa: E402

dev = 'cuda:0'
N, D = 10_000_001, int(sys.argv[1]) if len(sys.argv) > 1 else 128
U = int(os.environ.get('MB_U', '1024'))
torch.manual_seed(0)
W = torch.randn(N, D, device=dev) * 0.05
ue = torch.randn(U, D, device=dev)
# a history of 50 items per user (ascending columns), as Trainer.evaluate hands it over
H = 50
cols = torch.sort(torch.randint(1, N, (U, H), device=dev), dim=1).values.reshape(-1)
indptr = torch.arange(0, U * H + 1, H, device=dev, dtype=torch.int64)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)


def timed(fn, reps=3):
    fn(); torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        r = fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps, r


out = torch.empty(U, N, device=dev)
ms_plain, _ = timed(lambda: F_.fullsort_scores(ue, W, out=out))
del out
torch.cuda.empty_cache()
for kk in [int(x) for x in os.environ.get('MB_K', '').split(',') if x]:
    ms_k, _ = timed(lambda: F_.fullsort_topk(ue, W, None, kk, exclude_first_col=True))
    print(f'  (top-{kk} without a history mask: {ms_k:.2f} ms)')
if os.environ.get('MB_NOHIST'):
    ms_nh, _ = timed(lambda: F_.fullsort_topk(ue, W, None, 10, exclude_first_col=True))
    print(f'  (top-10 without a history mask: {ms_nh:.2f} ms)')
ms_topk, (tv, ti) = timed(lambda: F_.fullsort_topk(ue, W, None, 10, hist_indptr=indptr, hist_cols=cols, exclude_first_col=True))
# spot check of the lists against an fp64 product for 4 users
for u in (0, 1, U // 2, U - 1):
    sc = (ue[u].double() @ W.double().t()).float()
    sc[0] = -float('inf'); sc[cols[u * H:(u + 1) * H]] = -float('inf')
    want = torch.topk(sc, 10)
    assert torch.equal(torch.sort(want.indices).values, torch.sort(ti[u].long()).values) or float((want.values - tv[u]).abs().max()) < 1e-4, u
print(f'U={U} N={N} D={D}: plain {ms_plain:.2f} ms ({2.0*U*N*D/ms_plain/1e9:.1f} TFLOP/s)   mask+top-10 {ms_topk:.2f} ms ({2.0*U*N*D/ms_topk/1e9:.1f} TFLOP/s)   ratio {ms_topk/ms_plain:.3f}')
