"""Pointwise fused step (EMCDR default MF model) at C5 shapes: 2 x 1,048,576 rows per domain step (positives + one sampled
negative each, labels 1/0), D = 128, row-wise Adam.  Per-kernel HIP-event times and algorithmic bandwidth."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import recbole_cdr_amd  # noqa: F401
from recbole_cdr_amd import binding as B_
from recbole_cdr_amd.fused import FusedPointStep

dev = torch.device('cuda', 0)
nu, TOI, D, S = int(os.environ.get('NU', 50_000_001)), 10_000_000, 128, 1 << 20
ni, B = 1 + 2 * TOI, 2 * S
g = torch.Generator(device=dev); g.manual_seed(1)
U = torch.randn(nu, D, device=dev) * 0.01; I = torch.randn(ni, D, device=dev) * 0.01
st = FusedPointStep(U, I, B, loss='mse', opt='adam', reg_weight=0.01)
def batch():
    u = torch.randint(1, nu, (S,), device=dev, generator=g)
    p = torch.randint(1, 1 + TOI, (S,), device=dev, generator=g); n = torch.randint(1, 1 + TOI, (S,), device=dev, generator=g)
    return torch.cat([u, u]), torch.cat([p, n]), torch.cat([torch.ones(S, device=dev), torch.zeros(S, device=dev)])
bs = [batch() for _ in range(4)]
for i in range(3): st.step(*bs[i % 4])
B_.timing_enable(dev, 256)
torch.cuda.synchronize(); t0 = time.perf_counter()
for i in range(20): st.step(*bs[i % 4])
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20 * 1e3
tm = {}
for name, ms in B_.timing_collect(dev): tm.setdefault(name, []).append(ms)
uq_u, uq_i = int(torch.unique(bs[0][0]).numel()), int(torch.unique(bs[0][1]).numel())
alg = {'point_fwd_grad_kernel': B * (2 * 4 * D + 20) + B * 2 * 4 * D}
print(f'pointwise fused step: {dt:.3f} ms per domain step of {B} rows = {B / dt / 1e3:.1f} M rows/s', flush=True)
for k, v in tm.items():
    ms = sum(v) / len(v)
    extra = f'  {alg[k] / ms / 1e9:.2f} TB/s algorithmic' if k in alg else ''
    print(f'  {k}: {ms:.3f} ms{extra}')
ap = tm.get('rowwise_apply_kernel(users)', [])
if ap:
    mu, mi = sum(ap[0::2]) / len(ap[0::2]), sum(ap[1::2]) / len(ap[1::2])
    print(f'  apply users {mu:.3f} ms ({(B * (8 + 4 * D) + uq_u * 6 * 4 * D) / mu / 1e9:.2f} TB/s)  items {mi:.3f} ms ({(B * (8 + 4 * D) + uq_i * 6 * 4 * D) / mi / 1e9:.2f} TB/s)')
