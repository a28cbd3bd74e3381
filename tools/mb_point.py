"""Micro-benchmark of the pointwise fused gather-dot-loss forward (EMCDR-MF / CMF / BiTGCF path)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import recbole_cdr_amd
from recbole_cdr_amd import binding as B_
dev = 'cuda:0'
for (nu, ni, D, B) in ((6984, 3900, 64, 4096), (50_000_001, 20_000_001, 128, 1 << 21), (50_000_001, 20_000_001, 64, 1 << 22)):
    U = torch.randn(nu, D, device=dev) * 0.05; I = torch.randn(ni, D, device=dev) * 0.05
    u = torch.randint(1, nu, (B,), device=dev); i = torch.randint(1, ni, (B,), device=dev)
    y = (torch.rand(B, device=dev) < 0.5).float()
    out4 = torch.empty(4, device=dev); g = torch.empty(B, device=dev)
    for kind in (0, 1):
        call = lambda: B_.call('cdr_point_fwd', B_.ctx(dev), B_.stream(), kind, B_.f32(U), B_.f32(I), None, None, D, B_.i64(u), B_.i64(i), B_.f32(y), B, 0.01, B_.f32(out4), B_.f32(g), None)
        for _ in range(3): call()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): call()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
        print(f'kind={"MSE" if kind == 0 else "BCE"} nu={nu} D={D} B={B}: {ms*1e3:.1f} us  {B/ms/1e6:.2f} G rows/s  {B*(2*4*D+20)/ms/1e9:.3f} TB/s', flush=True)
    del U, I
