"""OVERLAP step (EMCDR map loss) at C5 table sizes: the two-launch distinct-id path against the general path; OB = 100 (reference
default) and 65,536; linear 128x128 and tanh-MLP 128-128-128 mappings.  6,144 B of table traffic per id (2 rows x 6 x 4D)."""
import os
import sys
import time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import recbole_cdr_amd  # noqa: E402
from recbole_cdr_amd import binding as B_, functional as F_  # noqa: E402
from recbole_cdr_amd.fused import FusedMapStep, RowwiseState, OPT_ADAM  # noqa: E402

dev = 'cuda:0'
NU = int(os.environ.get('NU', 50_000_001)); D = 128
S = torch.empty(NU, D, device=dev).normal_(0, 1e-3)
T = torch.empty(NU, D, device=dev).normal_(0, 1e-3)
ss, ts_ = RowwiseState(S, OPT_ADAM), RowwiseState(T, OPT_ADAM)
g = torch.Generator(device=dev).manual_seed(0)
for kind in ('linear', 'mlp'):
    if kind == 'linear':
        W = torch.nn.Parameter(torch.randn(D, D, device=dev) * 0.05)
        params, layers = [W], [(W, None, B_.ACT_NONE)]
        fn = lambda x: F_.linear(x, W, None, B_.ACT_NONE)
    else:
        W1, b1 = torch.nn.Parameter(torch.randn(128, D, device=dev) * 0.05), torch.nn.Parameter(torch.zeros(128, device=dev))
        W2, b2 = torch.nn.Parameter(torch.randn(D, 128, device=dev) * 0.05), torch.nn.Parameter(torch.zeros(D, device=dev))
        params, layers = [W1, b1, W2, b2], [(W1, b1, B_.ACT_TANH), (W2, b2, B_.ACT_NONE)]
        fn = lambda x: F_.linear(F_.linear(x, W1, b1, B_.ACT_TANH), W2, b2, B_.ACT_NONE)
    for OB in (100, 65536):
        perm = torch.randperm(NU - 1, device=dev, generator=g)[:OB * 8] + 1
        idxs = [perm[i * OB:(i + 1) * OB].view(-1, 1).contiguous() for i in range(8)]
        fm = FusedMapStep(S, T, fn, params, OB, opt='adam', lr=1e-3, layers=layers, source_state=ss, target_state=ts_)
        fm.capture(OB)
        for i in range(5):
            fm.replay(idxs[i % 8])
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for i in range(200):
            fm.replay(idxs[i % 8])
        torch.cuda.synchronize()
        print('%-6s OB %6d %-12s: %.4f ms per step (wall), one id copy + one graph launch' % (kind, OB, 'hipGraph', (time.perf_counter() - t0) / 200 * 1e3))
        for name, kw in (('general', {}), ('distinct-ids', {'unique': True})):
            for i in range(5):
                fm.step(idxs[i % 8], **kw)
            torch.cuda.synchronize()
            B_.timing_enable(dev, 4096)
            n = 100
            t0 = time.perf_counter()
            for i in range(n):
                fm.step(idxs[i % 8], **kw)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / n * 1e3
            ks = {}
            for nm, ms in B_.timing_collect(dev):
                ks.setdefault(nm, []).append(ms)
            B_.timing_enable(dev, 0)
            kt = float(np.mean(ks['map_step_kernel'])) if 'map_step_kernel' in ks else float('nan')
            byts = OB * 2 * 6 * 4 * D
            print('%-6s OB %6d %-12s: %.4f ms per step (wall) = %.1f M ids/s, %.2f TB/s on 6,144 B/id | map_step_kernel %.4f ms = %.2f TB/s'
                  % (kind, OB, name, dt, OB / dt / 1e3, byts / dt / 1e9, kt, byts / kt / 1e9 if kt == kt else 0))
