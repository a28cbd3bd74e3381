"""Per-positive pointwise step (fused.KMajorPointStep) against the per-row forms at C5 shapes: S = 1,048,576 positives + k = 1 sampled
negative each (2 M rows per domain step), D = 128, row-wise Adam.  ms per domain step of each form on the same batches."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import recbole_cdr_amd  # noqa: F401
from recbole_cdr_amd.fused import FusedPointStep, KMajorPointStep

dev = torch.device('cuda', 0)
nu, TOI, D, S, k = int(os.environ.get('NU', 50_000_001)), 10_000_000, 128, 1 << 20, int(os.environ.get('MB_K', 1))
ni = 1 + 2 * TOI
g = torch.Generator(device=dev); g.manual_seed(1)
U = torch.randn(nu, D, device=dev) * 0.01; I = torch.randn(ni, D, device=dev) * 0.01
def batch():
    u = torch.randint(1, nu, (S,), device=dev, generator=g)
    p = torch.randint(1, 1 + TOI, (S,), device=dev, generator=g); n = torch.randint(1, 1 + TOI, (S * k,), device=dev, generator=g)
    return u.repeat(1 + k), torch.cat([p, n]), torch.cat([torch.ones(S, device=dev), torch.zeros(S * k, device=dev)])
bs = [batch() for _ in range(4)]
for name, mk in (('per positive (KMajorPointStep)', lambda: KMajorPointStep(U, I, S, k=k, loss='mse', opt='adam', reg_weight=0.01)),
                 ('per row, one call (cdr_point_step_fused)', lambda: FusedPointStep(U, I, S * (1 + k), loss='mse', opt='adam', reg_weight=0.01)),
                 ('per row, two-pass', lambda: FusedPointStep(U, I, S * (1 + k), loss='mse', opt='adam', reg_weight=0.01, fuse_singles=False))):
    st = mk()
    for i in range(3): st.step(*bs[i % 4])
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(20): st.step(*bs[i % 4])
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20 * 1e3
    rows = S * (1 + k)
    print(f'{name:45s} {dt:7.3f} ms per domain step of {rows} rows = {rows / dt / 1e3:7.1f} M rows/s', flush=True)
    del st; torch.cuda.empty_cache()
# rows that are NOT tiled (every row its own user: arbitrary pointwise rows, what cdr_point_step_fused is for)
def batch_flat():
    n = S * (1 + k)
    return (torch.randint(1, nu, (n,), device=dev, generator=g), torch.randint(1, 1 + TOI, (n,), device=dev, generator=g),
            (torch.rand(n, device=dev, generator=g) < 0.5).float())
bf = [batch_flat() for _ in range(4)]
for name, fuse in (('untiled rows, one call (cdr_point_step_fused)', True), ('untiled rows, two-pass', False)):
    st = FusedPointStep(U, I, S * (1 + k), loss='mse', opt='adam', reg_weight=0.01, fuse_singles=fuse)
    for i in range(3): st.step(*bf[i % 4])
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(20): st.step(*bf[i % 4])
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20 * 1e3
    print(f'{name:45s} {dt:7.3f} ms per domain step of {S * (1 + k)} rows', flush=True)
    del st; torch.cuda.empty_cache()
