import os, sys, ctypes
sys.path.insert(0, '/root/repo')
import torch
import recbole_cdr_amd
from recbole_cdr_amd import binding as B_
dev = torch.device('cuda', 0)
def run(n, rows):
    ids = torch.randint(0, rows, (n,), device=dev)
    keys = torch.empty(n, device=dev, dtype=torch.int32); perm = torch.empty(n, device=dev, dtype=torch.int32)
    need = ctypes.c_size_t(0); B_.load().cdr_sort_workspace_bytes(n, rows, ctypes.byref(need))
    ws = torch.empty(need.value, device=dev, dtype=torch.uint8)
    ctxh = B_.ctx(dev)
    f = lambda: B_.call('cdr_sort_ids', ctxh, B_.stream(), B_.i64(ids), n, None, 0, rows, B_.raw(keys), B_.raw(perm), B_.raw(ws), ws.numel())
    for _ in range(3): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): f()
    e1.record(); torch.cuda.synchronize()
    print(f'sort n={n} rows={rows}: {e0.elapsed_time(e1)/50:.4f} ms', flush=True)
run(1 << 20, 50_000_001); run(2 << 20, 20_000_001); run(3 << 20, 1 << 27); run(3 << 20, 1 << 24); run(1<<20, 1<<24); run(1<<20, 1<<16)
