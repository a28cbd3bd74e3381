"""C5 sizes, both domains: the two domain steps of a bench step back to back on one stream vs side by side on two streams."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import recbole_cdr_amd  # noqa: F401
from recbole_cdr_amd.fused import FusedBPRStep
nu, ni, B, D = 50_000_001, 20_000_001, 1 << 20, 128
dev = torch.device('cuda', 0)
g = torch.Generator(device=dev); g.manual_seed(2022)
tabs = {k: torch.empty(r, D, device=dev).normal_(0, 0.01, generator=g) for k, r in (('su', nu), ('si', ni), ('tu', nu), ('ti', ni))}
steps = {'source': FusedBPRStep(tabs['su'], tabs['si'], B, opt='adam', reg_weight=0.01), 'target': FusedBPRStep(tabs['tu'], tabs['ti'], B, opt='adam', reg_weight=0.01)}
half = ni // 2
bs = [{d: (torch.randint(1, nu, (B,), device=dev, generator=g), torch.randint(lo, lo + half, (B,), device=dev, generator=g),
           torch.randint(lo, lo + half, (B,), device=dev, generator=g)) for d, lo in (('source', 1 + half), ('target', 1))} for _ in range(4)]
streams = {d: torch.cuda.Stream(device=dev) for d in steps}
def seq(i):
    for d in ('source', 'target'):
        steps[d].step(*bs[i % 4][d])
def par(i):
    cur = torch.cuda.current_stream()
    for d in ('source', 'target'):
        streams[d].wait_stream(cur)
        with torch.cuda.stream(streams[d]):
            steps[d].step(*bs[i % 4][d])
    for d in ('source', 'target'):
        cur.wait_stream(streams[d])
for name, fn in (('sequential', seq), ('two streams', par), ('sequential', seq), ('two streams', par)):
    for i in range(4):
        fn(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    N = 20
    for i in range(N):
        fn(i)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / N * 1e3
    print('%-12s %.3f ms per step (2 x %d triples) = %.3e interactions/s' % (name, dt, B, 2 * B / dt * 1e3), flush=True)
