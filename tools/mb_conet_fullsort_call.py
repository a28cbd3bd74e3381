"""Host-visible time of ONE CoNet.full_sort_predict call in evaluation mode (what recbole's evaluation loop pays per eval batch: one user at
the default eval_batch_size), back-to-back calls, C3's catalogue and a 1 M-item one: the general path (gather + Q contraction + scoring
kernel) against the few-users call (cdr_conet_fullsort_users: one launch, arguments packed once).
  python tools/mb_conet_fullsort_call.py"""
import json
import sys
import time

import torch

sys.path.insert(0, '.')
import recbole_cdr_amd  # noqa: F401,E402
from recbole_cdr_amd.data.synthetic import SyntheticCrossDomainDataset  # noqa: E402
from recbole_cdr_amd.model.cross_domain_recommender.conet import CoNet  # noqa: E402

dev = torch.device('cuda:0')
D, layers = 128, [64, 32, 16, 8]
res = {'D': D, 'tower': layers, 'cases': []}
for N in (18564, 1_000_001):
    ds = SyntheticCrossDomainDataset(OU=5983, TOU=2000, SOU=2000, OI=1, TOI=N - 1, SOI=1000, n_source_inter=2000, n_target_inter=2000)
    cfg = {'source_domain': {'NEG_PREFIX': 'neg_'}, 'target_domain': {'NEG_PREFIX': 'neg_'}, 'device': dev, 'embedding_size': D,
           'reg_weight': 0.01, 'mlp_hidden_size': layers}
    torch.manual_seed(2022)
    model = CoNet(cfg, ds).to(dev)
    model.eval()
    for U in (1, 4):
        inter = {model.TARGET_USER_ID: torch.arange(1, 1 + U, device=dev, dtype=torch.int64)}
        row = {'N': N, 'U': U}
        from recbole_cdr_amd import functional as F_
        F_.ConetFullsortFewUsers.MAX_PAIRS = 1 << 40                 # (time the one-launch call at every size; the product stops at 262,144 pairs)
        for name in ('few_users_one_launch', 'general_three_launches'):
            model.freeze_for_eval()                          # (drops the caches and opens the bracket CrossDomainTrainer.evaluate opens)
            with torch.no_grad():
                model.full_sort_predict({model.TARGET_USER_ID: torch.arange(1, 20, device=dev)})      # builds P (+ the packed call)
                if name.startswith('general'):
                    model.__dict__['_eval_few'] = None
                for _ in range(20):
                    model.full_sort_predict(inter)
                torch.cuda.synchronize()
                reps = 300 if N < 100000 else 60
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                t0 = time.perf_counter()
                e0.record()
                for _ in range(reps):
                    sc = model.full_sort_predict(inter)
                e1.record()
                t_host = (time.perf_counter() - t0) / reps
                torch.cuda.synchronize()
                row[name] = {'device_us_per_call': round(1e3 * e0.elapsed_time(e1) / reps, 2), 'host_enqueue_us_per_call': round(1e6 * t_host, 2)}
        res['cases'].append(row)
    del model
    torch.cuda.empty_cache()
print(json.dumps(res, indent=1))
