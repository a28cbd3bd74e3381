"""Fused BPR step at C5 shapes with Zipf(1.05) positive items (SURVEY 8d synthetic inputs (ii)): long duplicate segments
in the item apply.  Prints per-kernel HIP-event times for uniform vs Zipf positives."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import recbole_cdr_amd  # noqa: F401
from recbole_cdr_amd import binding as B_
from recbole_cdr_amd.fused import FusedBPRStep

dev = torch.device('cuda', 0)
nu, TOI, D, B = int(os.environ.get("NU", 20_000_001)), 10_000_000, 128, int(os.environ.get("MB_B", 1 << 20))
ni = 1 + 2 * TOI
g = torch.Generator(device=dev); g.manual_seed(1)
U = torch.randn(nu, D, device=dev) * 0.01; I = torch.randn(ni, D, device=dev) * 0.01
st = FusedBPRStep(U, I, B, opt='adam', reg_weight=0.01)
w = torch.arange(1, TOI + 1, device=dev, dtype=torch.float64).pow_(-1.05)
cdf = torch.cumsum(w, 0); cdf /= cdf[-1].clone()
popular_to_id = torch.randperm(TOI, device=dev, generator=g) + 1          # popularity rank -> item id
def batch(kind):
    u = torch.randint(1, nu, (B,), device=dev, generator=g)
    n = torch.randint(1, 1 + TOI, (B,), device=dev, generator=g)
    if kind == 'uniform':
        p = torch.randint(1, 1 + TOI, (B,), device=dev, generator=g)
    else:
        r = torch.searchsorted(cdf, torch.rand(B, device=dev, generator=g, dtype=torch.float64)).clamp_(max=TOI - 1)
        p = popular_to_id[r]
    return u, p, n
for kind in ('uniform', 'zipf', 'uniform', 'zipf'):
    bs = [batch(kind) for _ in range(4)]
    _, cnt = torch.unique(bs[0][1], return_counts=True)
    for i in range(3): st.step(*bs[i % 4])
    B_.timing_enable(dev, 256)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(10): st.step(*bs[i % 4])
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10 * 1e3
    tm = {}
    for name, ms in B_.timing_collect(dev): tm.setdefault(name, []).append(ms)
    B_.timing_enable(dev, 0)
    print(f'{kind:8s} step {dt:7.3f} ms | distinct pos {cnt.numel()} longest segment {int(cnt.max())} | ' +
          ' '.join(f'{k}={sum(v)/len(v):.3f}' for k, v in tm.items()), flush=True)
