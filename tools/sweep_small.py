"""Reference-default batch sizes at the C5 table sizes through the per-positive step (KMajorBPRStep), eager and as one
hipGraph: ms per domain step (wall clock over many steps, ids already on the device) and the device time of the graph."""
import os
import sys
import time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import recbole_cdr_amd  # noqa: E402
from recbole_cdr_amd.fused import KMajorBPRStep, FusedBPRStep  # noqa: E402

dev = 'cuda:0'
NU = int(os.environ.get('NU', 50_000_001)); NI = int(os.environ.get('NI', 20_000_001)); D = 128
U = torch.empty(NU, D, device=dev).normal_(0, 1e-3)
I = torch.empty(NI, D, device=dev).normal_(0, 1e-3)
g = torch.Generator(device=dev).manual_seed(0)
from recbole_cdr_amd.fused import RowwiseState, OPT_ADAM  # noqa: E402
us, its = RowwiseState(U, OPT_ADAM), RowwiseState(I, OPT_ADAM)      # one set of moments (72 GB) shared by the three step objects
ONLY = os.environ.get('ONLY')
for rows, k in ((2048, 1), (2048, 4), (8192, 1), (8192, 4)):
    if ONLY and ONLY != '%d,%d' % (rows, k):
        continue
    S = rows // k
    batches = [(torch.randint(1, NU, (S,), device=dev, generator=g), torch.randint(1, 10_000_001, (S,), device=dev, generator=g),
                torch.randint(1, 10_000_001, (rows,), device=dev, generator=g)) for _ in range(8)]
    old = FusedBPRStep(U, I, rows, opt='adam', lr=1e-3, reg_weight=0.01, user_state=us, item_state=its)
    new = KMajorBPRStep(U, I, S, k=k, opt='adam', lr=1e-3, reg_weight=0.01, user_state=us, item_state=its)
    gr = KMajorBPRStep(U, I, S, k=k, opt='adam', lr=1e-3, reg_weight=0.01, user_state=us, item_state=its)
    if not gr.small:
        print('rows %5d k %d : FusedBPRStep eager %.4f ms | k-major eager (radix sort) %.4f ms' % (rows, k, 0, 0)); continue
    gr.capture(S)
    def run(fn, n=300):
        for i in range(20):
            fn(batches[i % 8])
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for i in range(n):
            fn(batches[i % 8])
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3
    t_old = run(lambda b: old.step(b[0].repeat(k), b[1].repeat(k), b[2]))
    t_new = run(lambda b: new.step(*b))
    t_gr = run(lambda b: gr.replay(*b))
    t_in = run(lambda b: gr.replay())
    # device time of the graph alone (no id copies): replay back to back
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for i in range(300):
        gr._graph.replay()
    e1.record(); torch.cuda.synchronize()
    t_dev = e0.elapsed_time(e1) / 300
    print('rows %5d k %d : FusedBPRStep eager %.4f ms | k-major eager %.4f ms | k-major hipGraph %.4f ms (graph alone, device %.4f ms) '
          '| ids written in place %.4f ms | %.1f M rows/s' % (rows, k, t_old, t_new, t_gr, t_dev, t_in, rows / t_gr / 1e3))
