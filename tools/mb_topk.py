"""Fused scoring + mask + top-k (cdr_fullsort_topk_f32) vs scoring into a [U, N] matrix followed by torch.topk,
N = 10,000,001 items (BASELINE C5 target slab)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import recbole_cdr_amd  # noqa: F401
from recbole_cdr_amd import functional as F_

dev = torch.device('cuda', 0)
N, D, k = 10_000_001, int(sys.argv[1]) if len(sys.argv) > 1 else 128, 10
items = torch.randn(N, D, device=dev) * 0.05

def timeit(fn, reps):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps

for U in [int(x) for x in os.environ.get('MB_U', '1,8,64,256,1024').split(',')]:
    ue = torch.randn(U, D, device=dev)
    hist_n = 50
    cols = torch.sort(torch.randint(1, N, (U, hist_n), device=dev), dim=1).values
    indptr = torch.arange(0, U + 1, device=dev, dtype=torch.int64) * hist_n
    hist = cols.reshape(-1).contiguous()
    reps = 3 if U >= 256 else 10
    t_f = timeit(lambda: F_.fullsort_topk(ue, items, None, k=k, hist_indptr=indptr, hist_cols=hist), reps)
    out = torch.empty(U, N, device=dev)
    rows = torch.arange(U, device=dev).repeat_interleave(hist_n)
    def unfused():
        F_.fullsort_scores(ue, items, out=out)
        out[:, 0] = -float('inf'); out[rows, hist] = -float('inf')
        return torch.topk(out, k, dim=1)
    t_s = timeit(lambda: F_.fullsort_scores(ue, items, out=out), reps)
    t_u = timeit(unfused, reps)
    a, b = F_.fullsort_topk(ue, items, None, k=k, hist_indptr=indptr, hist_cols=hist), unfused()
    ok = torch.equal(a[0], b.values)
    print(f'U={U:5d} D={D}: fused top-{k} {t_f:8.3f} ms ({U * N / t_f / 1e6:8.1f} G items/s, {2.0 * U * N * D / t_f / 1e9:6.1f} TFLOP/s) | '
          f'scores only {t_s:8.3f} ms | scores + mask + torch.topk {t_u:8.3f} ms | values equal {ok}', flush=True)
    del out
