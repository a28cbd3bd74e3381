"""SURVEY 8d sweep at the C5 table sizes: fused BPR step (row-wise Adam) for batch rows in {2,048, 65,536, 1,048,576} x k in {1, 4}
(k negatives per positive: the S = rows / k positives are repeated k times, negatives k-major -- recbole's pairwise layout) and
the OVERLAP step for OB in {100, 65,536}.  Wall-clock per step and rows/s; the small batches are launch-bound."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import recbole_cdr_amd  # noqa: F401
from recbole_cdr_amd import functional as F_, binding as B_
from recbole_cdr_amd.fused import FusedBPRStep, FusedMapStep

dev = torch.device('cuda', 0)
nu, TOI, D = int(os.environ.get('NU', 50_000_001)), 10_000_000, 128
g = torch.Generator(device=dev); g.manual_seed(2022)
U = torch.randn(nu, D, device=dev) * 0.01; I = torch.randn(1 + 2 * TOI, D, device=dev) * 0.01
U2 = torch.randn(nu, D, device=dev) * 0.01
st = FusedBPRStep(U, I, 1 << 20, opt='adam', reg_weight=0.01)
def timeit(fn, reps):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3
for rows in (2048, 65536, 1 << 20):
    for k in (1, 4):
        S = rows // k
        u = torch.randint(1, nu, (S,), device=dev, generator=g).repeat(k)
        p = torch.randint(1, 1 + TOI, (S,), device=dev, generator=g).repeat(k)
        n = torch.randint(1, 1 + TOI, (S * k,), device=dev, generator=g)
        ms = timeit(lambda: st.step(u, p, n), 200 if rows < (1 << 20) else 30)
        print(f'BPR step   rows={S * k:8d} k={k}: {ms:8.4f} ms  {S * k / ms / 1e3:9.2f} M rows/s  ({S / ms / 1e3:8.2f} M positives/s)', flush=True)
W = torch.nn.Parameter(torch.randn(D, D, device=dev) * 0.05)
fm = FusedMapStep(U, U2, lambda x: F_.linear(x, W, None, B_.ACT_NONE), [W], 65536, source_state=st.ustate)
for OB in (100, 65536):
    idx = torch.randperm(nu - 1, device=dev)[:OB].add(1).reshape(-1, 1)
    ms = timeit(lambda: fm.step(idx), 100)
    print(f'OVERLAP step OB={OB:6d}: {ms:8.4f} ms  {OB / ms / 1e3:9.3f} M ids/s', flush=True)
