"""Local cost of the dimension-sharded POINTWISE step's second half (after the all-reduce of the partial dots) at world 1 on one GPU:
the forward-and-update pass fed with the given dots (cdr_point_step_from_dot) against the two-pass form
(cdr_point_grad_from_dot -> 2 x cdr_rowwise_apply).
  python tools/mb_dimshard_point.py [rows_u rows_i B D]"""
import json
import sys

import torch

sys.path.insert(0, '.')
import recbole_cdr_amd  # noqa: F401,E402
from recbole_cdr_amd.dimshard import DimShardedPointStep, NativePointDimOps  # noqa: E402

nu, ni, B, D = (int(x) for x in sys.argv[1:5]) if len(sys.argv) >= 5 else (20_000_000, 5_000_000, 1 << 21, 64)
dev = torch.device('cuda:0')
res = {'rows_u': nu, 'rows_i': ni, 'B': B, 'D': D, 'runs': []}
for fused in (True, False):
    torch.manual_seed(0)
    U = torch.randn(nu, D, device=dev) * 0.01
    I = torch.randn(ni, D, device=dev) * 0.01
    ops = NativePointDimOps(U, I, B, loss='mse', opt='adam', lr=1e-3, reg_weight=1e-3, fuse_singles=fused)
    step = DimShardedPointStep(U, I, B, ops=ops)
    batches = [(torch.randint(1, nu, (B,), device=dev), torch.randint(1, ni, (B,), device=dev), (torch.rand(B, device=dev) < 0.5).float())
               for _ in range(4)]
    for b in batches[:2]:
        step.step(*b)
    torch.cuda.synchronize()
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    n_it = 10
    for it in range(n_it):
        step.step(*batches[it % 4])
    e.record()
    torch.cuda.synchronize()
    res['runs'].append({'second_half': 'one pass (cdr_point_step_from_dot)' if fused else 'two passes (grad_from_dot + rowwise_apply x 2)',
                        'ms_per_step': round(a.elapsed_time(e) / n_it, 3), 'loss': float(step.out[0])})
    del U, I, step, ops, batches
    torch.cuda.empty_cache()
res['speedup'] = round(res['runs'][1]['ms_per_step'] / res['runs'][0]['ms_per_step'], 2)
print(json.dumps(res, indent=1))
