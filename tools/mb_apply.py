"""Micro-benchmark of cdr_rowwise_apply alone at C5 shapes (sorted once, applied repeatedly): users (n = B over NU rows)
and items (n = 2B over 20 M rows, signed).  CDR_LIB_PATH selects the library build for A/B runs."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import recbole_cdr_amd  # noqa: F401
from recbole_cdr_amd import binding as B_
from recbole_cdr_amd.fused import FusedBPRStep

dev = torch.device('cuda', 0)
nu, ni, D, B = int(os.environ.get('NU', 20_000_001)), 20_000_001, 128, 1 << 20
g = torch.Generator(device=dev); g.manual_seed(1)
U = torch.randn(nu, D, device=dev) * 0.01; I = torch.randn(ni, D, device=dev) * 0.01
st = FusedBPRStep(U, I, B, opt='adam', reg_weight=0.01)
u = torch.randint(1, nu, (B,), device=dev, generator=g)
p = torch.randint(1, 10_000_001, (B,), device=dev, generator=g); n = torch.randint(1, 10_000_001, (B,), device=dev, generator=g)
st.step(u, p, n)
ctxh = B_.ctx(dev)
def timeit(fn, reps=30):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(reps):
        e0.record(); fn(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    ts.sort(); return ts[len(ts) // 2], ts[0]
import ctypes
B_.call('cdr_sort_ids_two_tables', ctxh, B_.stream(), B_.i64(u), B, nu, B_.i64(p), B, B_.i64(n), B, ni, B_.raw(st.keys), B_.raw(st.perm),
        ctypes.byref(st._key_base), B_.raw(st.ws), st.ws_bytes)
mu = timeit(lambda: st._apply(ctxh, st.ustate, st.keys[:B], st.perm[:B], B, st.GU, B, B, st.out6[4:5], 0))
mi = timeit(lambda: st._apply(ctxh, st.istate, st.keys[B:3 * B], st.perm[B:3 * B], 2 * B, st.GP, B, B, st.out6[5:6], st._key_base.value))
print(f'{os.environ.get("CDR_LIB_PATH", "default"):24s} apply users median {mu[0]:.4f} min {mu[1]:.4f} ms | items median {mi[0]:.4f} min {mi[1]:.4f} ms', flush=True)
