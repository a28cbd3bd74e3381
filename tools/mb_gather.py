"""Micro-benchmark of the fused gather-dot-loss forward at C5-like shapes (HIP-event timed)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import recbole_cdr_amd
from recbole_cdr_amd import functional as F_, binding as B_

dev = 'cuda:0'
def run(nu, ni, D, B, iters=20):
    U = torch.randn(nu, D, device=dev) * 0.05
    I = torch.randn(ni, D, device=dev) * 0.05
    u = torch.randint(1, nu, (B,), device=dev); p = torch.randint(1, ni, (B,), device=dev); n = torch.randint(1, ni, (B,), device=dev)
    out4 = torch.empty(4, device=dev); g = torch.empty(B, device=dev)
    def call():
        B_.call('cdr_bpr_fwd', B_.ctx(dev), B_.stream(), B_.f32(U), B_.f32(I), D, B_.i64(u), B_.i64(p), B_.i64(n), B, 1e-10, 0.01, B_.f32(out4), B_.f32(g))
    for _ in range(3): call()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): call()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    bytes_ = B * (3 * 4 * D + 24)
    print(f'nu={nu} ni={ni} D={D} B={B}: {ms*1e3:.1f} us/launch  {B/ms/1e6:.2f} G triples/s  {bytes_/ms/1e9:.3f} TB/s algorithmic', flush=True)
    del U, I

if __name__ == '__main__':
    run(6984, 3900, 64, 2048)
    run(1_000_000, 1_000_000, 128, 65536)
    run(1_000_000, 1_000_000, 128, 1 << 20)
    run(10_000_000, 10_000_000, 128, 1 << 20)
    run(50_000_001, 20_000_001, 128, 1 << 20)
    run(50_000_001, 20_000_001, 128, 1 << 22)
    run(50_000_001, 20_000_001, 64, 1 << 22)
