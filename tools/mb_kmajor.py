"""C5 table sizes, large batches: the per-positive step (KMajorBPRStep) against the per-triple step (FusedBPRStep), per-kernel
HIP-event times and ms per domain step.  rows = S * k triples per domain step."""
import os
import sys
import time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import recbole_cdr_amd  # noqa: E402
from recbole_cdr_amd import binding as B_  # noqa: E402
from recbole_cdr_amd.fused import KMajorBPRStep, FusedBPRStep, RowwiseState, OPT_ADAM  # noqa: E402

dev = 'cuda:0'
NU = int(os.environ.get('NU', 50_000_001)); NI = int(os.environ.get('NI', 20_000_001)); D = 128
ROWS = int(os.environ.get('ROWS', 1 << 20))
U = torch.empty(NU, D, device=dev).normal_(0, 1e-3)
I = torch.empty(NI, D, device=dev).normal_(0, 1e-3)
us, its = RowwiseState(U, OPT_ADAM), RowwiseState(I, OPT_ADAM)
g = torch.Generator(device=dev).manual_seed(0)
for k in (1, 4):
    S = ROWS // k
    batches = [(torch.randint(1, NU, (S,), device=dev, generator=g), torch.randint(1, 10_000_001, (S,), device=dev, generator=g),
                torch.randint(1, 10_000_001, (ROWS,), device=dev, generator=g)) for _ in range(4)]
    tiled = [(b[0].repeat(k), b[1].repeat(k), b[2]) for b in batches]
    old = FusedBPRStep(U, I, ROWS, opt='adam', lr=1e-3, reg_weight=0.01, user_state=us, item_state=its)
    new = KMajorBPRStep(U, I, S, k=k, opt='adam', lr=1e-3, reg_weight=0.01, user_state=us, item_state=its)
    rec = KMajorBPRStep(U, I, S, k=k, opt='adam', lr=1e-3, reg_weight=0.01, user_state=us, item_state=its, fuse_singles=False)   # round 2's form
    for name, fn, data in (('per-triple', lambda b: old.step(*b), tiled), ('per-positive', lambda b: new.step(*b), batches),
                           ('per-positive (records, r02)', lambda b: rec.step(*b), batches)):
        for i in range(5):
            fn(data[i % 4])
        torch.cuda.synchronize()
        B_.timing_enable(dev, 4096)
        t0 = time.perf_counter()
        n = 30
        for i in range(n):
            fn(data[i % 4])
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n * 1e3
        acc = {}
        for nm, ms in B_.timing_collect(dev):
            acc.setdefault(nm, []).append(ms)
        B_.timing_enable(dev, 0)
        ks = ', '.join('%s %.3f' % (nm.replace('_kernel', ''), float(np.mean(v))) for nm, v in acc.items())
        print('k=%d rows=%d %-27s: %.3f ms per domain step = %.0f M rows/s | %s' % (k, ROWS, name, dt, ROWS / dt / 1e3, ks))
    del old, new, rec
