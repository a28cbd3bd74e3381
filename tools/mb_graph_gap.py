#!/usr/bin/env python3
"""How long the GPU idles between two hipGraph launches (why CrossDomainTrainer replays 8 steps per launch on a device loader): a graph
of K empty-ish kernels (cdr_inc_i64 on one counter) replayed back to back, K = 1, 5, 40.  per-replay time = K x kernel time + gap.
Usage on an MI355X: python tools/mb_graph_gap.py"""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import recbole_cdr_amd  # noqa: F401,E402
from recbole_cdr_amd import binding as B_  # noqa: E402

DEV = 'cuda:0'
cnt = torch.zeros(1, device=DEV, dtype=torch.int64)
res = {}
for K in (1, 5, 40):
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        B_.call('cdr_inc_i64', B_.stream(), B_.i64(cnt))
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with B_.capturing(g, side):
        for _ in range(K):
            B_.call('cdr_inc_i64', B_.stream(), B_.i64(cnt))
    for _ in range(20):
        g.replay()
    torch.cuda.synchronize()
    n = 400
    t0 = time.perf_counter()
    for _ in range(n):
        g.replay()
    torch.cuda.synchronize()
    res[K] = (time.perf_counter() - t0) / n * 1e6
per_kernel = (res[40] - res[5]) / 35.0
out = {'us_per_replay': res, 'us_per_kernel_inside_a_graph': per_kernel, 'us_between_graph_launches': {K: v - K * per_kernel for K, v in res.items()},
       'what': 'graphs of K one-thread kernels replayed back to back (400 replays); the second figure is the slope, the third the intercept'}
print(json.dumps(out))
