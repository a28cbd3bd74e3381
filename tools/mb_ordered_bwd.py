"""Micro-benchmark of the three routes to a drop-in loss's dense gradients (atomic / sorted / ordered: see tests/test_gpu_ordered.py)
at the reference's batch sizes: forward + backward of BPRGatherLoss and TwoDomainPointLoss captured in a hipGraph and replayed, so that
the figure is device time per step, not Python.  Usage: python tools/mb_ordered_bwd.py > profiles/rNN_mb_ordered_bwd.txt"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import recbole_cdr_amd  # noqa: E402,F401
from recbole_cdr_amd import functional as F_, binding as B_  # noqa: E402

DEV = torch.device('cuda:0')


def zipf_ids(n, hi, a, gen):
    w = 1.0 / torch.arange(1, hi + 1, dtype=torch.float64) ** a
    return torch.multinomial(w / w.sum(), n, replacement=True, generator=gen)


def timed(step, reps=int(os.environ.get("MB_REPS", "300"))):
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):                 # the library keeps one context per stream: warm it up on the capture stream
        for _ in range(3):
            step()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with B_.capturing(g, st):
        step()
    for _ in range(20):
        g.replay()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3     # us


def main():
    gen = torch.Generator().manual_seed(1)
    print('forward + backward per replayed step, us (MI355X); routes: atomic (default) | sorted (set_deterministic, ordered off) | ordered (set_deterministic; cap lifted to 16,384 for the table)')
    for name, nu, ni, D in (('C2-like (ml-1m -> ml-100k, D = 64)', 6041, 3707, 64), ('D = 128, 50 k x 20 k rows', 50000, 20000, 128)):
        U = (torch.randn(nu, D, generator=gen) * 0.1).to(DEV).requires_grad_(True)
        I = (torch.randn(ni, D, generator=gen) * 0.1).to(DEV).requires_grad_(True)
        for n in ((2048,) if os.environ.get('MB_QUICK') else (2048, 4096, 8192)):
            for dist in ('uniform', 'zipf1.05'):
                if dist == 'uniform':
                    u, p, q = (torch.randint(0, h, (n,), generator=gen).to(DEV) for h in (nu, ni, ni))
                else:
                    u, p, q = (zipf_ids(n, h, 1.05, gen).to(DEV) for h in (nu, ni, ni))
                y = (torch.rand(n, generator=gen) < 0.5).float().to(DEV)
                hot = int(torch.bincount(p).max())

                def bpr():
                    U.grad = I.grad = None
                    F_.BPRGatherLoss.apply(U, I, u, p, q, 1e-10, 0.01).sum().backward()

                def pair():
                    U.grad = I.grad = None
                    F_.TwoDomainPointLoss.apply(B_.CDR_LOSS_BCE, U, I, u, p, y, 0.01, u.flip(0), q, 1 - y, 0.01, 0.5)[0].sum().backward()

                for lname, fn in (('BPRGatherLoss', bpr), ('TwoDomainPointLoss', pair)):
                    out = []
                    try:
                        F_.set_ordered_backward(False)
                        F_.set_deterministic(False)
                        out.append(timed(fn))
                        F_.set_deterministic(True)
                        out.append(timed(fn))
                        F_.set_ordered_backward(True, max_entries=16384)
                        out.append(timed(fn))
                    finally:
                        F_.set_deterministic(False)
                        F_.set_ordered_backward(True)
                    print(f'{name:38s} n={n:5d} {dist:9s} hottest item x{hot:4d}  {lname:19s} atomic {out[0]:7.2f}  sorted {out[1]:7.2f}  ordered {out[2]:7.2f}')


if __name__ == '__main__':
    main()
