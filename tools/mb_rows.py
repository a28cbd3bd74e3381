"""cdr_gather_rows at exchange sizes: 2 M rows of 512 B out of a 20 M-row table (the owner-side gather of the sharded step)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import recbole_cdr_amd  # noqa: F401
from recbole_cdr_amd import binding as B_
dev = torch.device('cuda', 0)
for rows, n, D in ((20_000_001, 2 << 20, 128), (20_000_001, 2 << 20, 64), (50_000_001, 65536, 128)):
    tab = torch.randn(rows, D, device=dev); ids = torch.randint(0, rows, (n,), device=dev); out = torch.empty(n, D, device=dev)
    f = lambda: B_.call('cdr_gather_rows', B_.stream(), B_.f32(tab), D, B_.i64(ids), n, B_.f32(out))
    for _ in range(3): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): f()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    assert torch.equal(out, tab[ids])
    print(f'gather_rows rows={rows} n={n} D={D}: {ms:.4f} ms  {(2 * 4 * D + 8) * n / ms / 1e9:.2f} TB/s', flush=True)
    del tab, out
