#!/usr/bin/env python3
"""Measured random-row gather bandwidth by working-set size: the roof the cache-resident configurations
(BASELINE C1 / C2: a few MB of tables; C4: 43 MB) are bounded by is the L2 / Infinity-Cache gather rate, not the 8 TB/s of HBM.
One pure gather-reduce launch (cdr_embloss_fwd: the squared norms of B user rows and B item rows -- reads 2 B rows, writes 3 floats)
over two tables of the given total footprint, B random ids each, D = 64 and 128.  bytes = 2 B D 4 per launch.
Usage on an MI355X: python tools/mb_cache_gather.py  -> one JSON line per (D, footprint)"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import recbole_cdr_amd  # noqa: F401,E402
from recbole_cdr_amd import binding as B_  # noqa: E402

DEV = 'cuda:0'


def measure(D, footprint_bytes, B=1 << 20, reps=50):
    rows = max(footprint_bytes // (2 * 4 * D), 16)
    g = torch.Generator(device=DEV); g.manual_seed(1)
    U = torch.randn(rows, D, device=DEV, generator=g)
    I = torch.randn(rows, D, device=DEV, generator=g)
    u = torch.randint(0, rows, (B,), device=DEV, generator=g)
    i = torch.randint(0, rows, (B,), device=DEV, generator=g)
    out3 = torch.empty(3, device=DEV)
    call = lambda: B_.call('cdr_embloss_fwd', B_.ctx(DEV), B_.stream(), B_.f32(U), B_.f32(I), D, B_.i64(u), B_.i64(i), B, B_.f32(out3))
    for _ in range(5):
        call()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        call()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    byts = 2.0 * B * D * 4
    # the same working set streamed through the cache level it sits in (a sum over it): upper bound for any access pattern at that footprint
    flat = torch.cat([U.reshape(-1), I.reshape(-1)])
    for _ in range(3):
        flat.sum()
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        flat.sum()
    e1.record(); torch.cuda.synchronize()
    sms = e0.elapsed_time(e1) / reps
    return {'D': D, 'footprint_MB': 2 * rows * D * 4 / 1e6, 'rows_per_table': rows, 'gathered_rows_per_launch': 2 * B, 'gather_ms': ms,
            'random_row_gather_GBps': byts / (ms * 1e-3) / 1e9, 'stream_sum_ms': sms, 'stream_GBps': flat.numel() * 4.0 / (sms * 1e-3) / 1e9}


if __name__ == '__main__':
    for D in (64, 128):
        for mb in (1, 2, 3, 8, 16, 43, 128, 512, 4096, 32768):
            print(json.dumps(measure(D, mb * 1000 * 1000)), flush=True)
