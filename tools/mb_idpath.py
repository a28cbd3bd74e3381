"""Round 6: the fused BPR step's id work on medium batches -- the sorted path (make_keys + rocPRIM radix sort of 3 B pairs + occurrence flags) against
the count path (one counter per table row; only duplicate occurrences are sorted) at the C5 table sizes, uniform and Zipf(1.05) positives, eager
and replayed as a hipGraph.  One MI355X.  python tools/mb_idpath.py [users items D]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch

import recbole_cdr_amd  # noqa: F401
from recbole_cdr_amd import binding as B_
from recbole_cdr_amd.fused import FusedBPRStep, RowwiseState

dev = torch.device('cuda:0')
nu, ni, D = (int(x) for x in sys.argv[1:4]) if len(sys.argv) >= 4 else (50_000_000, 10_000_000, 128)
gen = torch.Generator(device=dev); gen.manual_seed(2022)
U = torch.empty(nu, D, device=dev).normal_(0, 0.01, generator=gen)
I = torch.empty(ni, D, device=dev).normal_(0, 0.01, generator=gen)
us, its = RowwiseState(U, 1), RowwiseState(I, 1)


def zipf(n):
    r = torch.rand(n, device=dev, generator=gen, dtype=torch.float64)
    a = 1.05
    x = ((float(ni - 1) ** (1 - a) - 1) * r + 1) ** (1 / (1 - a))
    return x.long().clamp_(1, ni - 1)


def timed(fn, batches, n):
    for b in batches:
        fn(*b)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n):
        fn(*batches[i % len(batches)])
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


res = {'users': nu, 'items': ni, 'D': D, 'cases': []}
for B in (32768, 65536, 131072):
    for dist in ('uniform', 'zipf(1.05) positives'):
        batches = [(torch.randint(1, nu, (B,), device=dev, generator=gen), zipf(B) if dist != 'uniform' else torch.randint(1, ni, (B,), device=dev, generator=gen),
                    torch.randint(1, ni, (B,), device=dev, generator=gen)) for _ in range(4)]
        row = {'B': B, 'ids': dist}
        for path in ('sort', 'count', 'auto'):
            st = FusedBPRStep(U, I, B, opt='adam', reg_weight=0.01, user_state=us, item_state=its, id_path=path)
            row[path + '_eager_ms'] = round(timed(st.step, batches, 60), 4)
            if path == 'auto':
                row['auto_settled_on'] = 'count' if st._use_count else 'sort'
            side = torch.cuda.Stream(device=dev)
            with torch.cuda.stream(side):
                st.step(*batches[0])
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with B_.capturing(g, side):
                st.step(*batches[0])

            def rep(*_a):
                g.replay(); st.replayed()
            row[path + '_replayed_ms'] = round(timed(rep, batches, 60), 4)
            row[path + '_dup_occurrences'] = int(st.heads[2])
            del g, st
        row['frac_of_hbm_peak_best_replayed'] = round(B * 9216 / (min(row['sort_replayed_ms'], row['count_replayed_ms']) * 1e-3) / 8e12, 3)
        res['cases'].append(row)
        print(row, file=sys.stderr)
print(json.dumps(res, indent=1))
