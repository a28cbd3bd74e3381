"""Local cost of the dimension-sharded BPR step at a simulated world size on ONE GPU: tables [rows, D/G], global batch
G x B (what every rank walks), no collectives.  Compared with the single-GPU fused step on [rows, D] and batch B.
  python tools/mb_dimshard.py [rows_u rows_i B D]"""
import json
import sys

import torch

sys.path.insert(0, '.')
import recbole_cdr_amd  # noqa: F401,E402
from recbole_cdr_amd import binding as B_  # noqa: E402
from recbole_cdr_amd.dimshard import DimShardedBPRStep  # noqa: E402
from recbole_cdr_amd.fused import FusedBPRStep  # noqa: E402

nu, ni, B, D = (int(x) for x in sys.argv[1:5]) if len(sys.argv) >= 5 else (50_000_000, 10_000_000, 1 << 20, 128)
dev = torch.device('cuda:0')
res = {'rows_u': nu, 'rows_i': ni, 'B_per_rank': B, 'D': D, 'runs': []}
# (G, column divisor, batch multiplier): the plain dimension layout at G ranks (every rank: D / G columns, G x B triples per DOMAIN, two
# domain steps per step) and the domain-groups form at 8 ranks (a rank serves ONE domain: D / 4 columns, 4 x 2B triples, one domain step per step)
SHAPES = [(1, 1, 1, 'dim'), (2, 2, 2, 'dim'), (4, 4, 4, 'dim'), (8, 8, 8, 'dim'), (8, 4, 8, 'dim-groups (one domain per rank: per STEP, not per domain step)')]
for G, cdiv, bmul, label in SHAPES:
    Ds = D // cdiv
    U = torch.randn(nu, Ds, device=dev) * 0.01
    I = torch.randn(ni, Ds, device=dev) * 0.01
    Bg = bmul * B
    step = DimShardedBPRStep(U, I, Bg, opt='adam', lr=1e-3, reg_weight=1e-3) if G > 1 else FusedBPRStep(U, I, Bg, opt='adam', lr=1e-3, reg_weight=1e-3)
    batches = [(torch.randint(1, nu, (Bg,), device=dev), torch.randint(1, ni, (Bg,), device=dev), torch.randint(1, ni, (Bg,), device=dev))
               for _ in range(4)]
    for b in batches[:2]:
        step.step(*b)
    torch.cuda.synchronize()
    ctx = B_.ctx(dev)
    B_.timing_enable(dev, 256)
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    n_it = 8
    for it in range(n_it):
        step.step(*batches[it % 4])
    e.record()
    torch.cuda.synchronize()
    ms = a.elapsed_time(e) / n_it
    per = {}
    for name, t in B_.timing_collect(dev):
        per.setdefault(name, []).append(t)
    B_.timing_enable(dev, 0)
    res['runs'].append({'G': G, 'layout': label, 'Ds': Ds, 'global_batch': Bg, 'ms_per_domain_step': round(ms, 3),
                        'vs_G1': None, 'kernels_ms': {k: round(sum(v) / len(v), 3) for k, v in per.items()}})
    del U, I, step, batches
    torch.cuda.empty_cache()
base = res['runs'][0]['ms_per_domain_step']
for r in res['runs']:
    r['vs_G1'] = round(r['ms_per_domain_step'] / base, 2)
    r['local_efficiency'] = round(base / r['ms_per_domain_step'], 2)
print(json.dumps(res, indent=1))
