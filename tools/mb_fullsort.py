"""Micro-benchmark of full-sort scoring at C5 size for several U (HIP-event timed)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import recbole_cdr_amd
from recbole_cdr_amd import functional as F_
dev = 'cuda:0'
N, D = 10_000_001, int(sys.argv[1]) if len(sys.argv) > 1 else 128
W = torch.randn(N, D, device=dev) * 0.05
for U in [int(x) for x in os.environ.get('MB_U', '1,4,16,32,64,128,256,1024').split(',')]:
    ue = torch.randn(U, D, device=dev)
    out = torch.empty(U, N, device=dev)
    F_.fullsort_scores(ue, W, out=out); torch.cuda.synchronize()
    reps = 5 if U <= 64 else 2
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): F_.fullsort_scores(ue, W, out=out)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    byts = 4.0 * N * D + 4.0 * U * N
    idx = torch.randint(0, N, (32,), device=dev)
    err = (out[:, idx] - (ue.double() @ W[idx].double().t()).float()).abs().max().item()
    print(f'U={U:5d} D={D}: {ms:8.3f} ms  {U*N/ms/1e6:9.1f} G items/s  {byts/ms/1e9:6.3f} TB/s  {2.0*U*N*D/ms/1e9:7.2f} TFLOP/s  maxerr {err:.2e}', flush=True)
    del out
