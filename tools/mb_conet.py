"""Micro-benchmark of the fused CoNet tower kernels at BASELINE C3's shape: HIP-event time per kernel (recorded in the
library on the launch stream) and FLOP rates.  `CDR_CONET_PROF=1` with a -DCDR_CONET_PROF build also prints block 0's
phase stamps (wall_clock64, 10 ns ticks)."""
import ctypes
import os
import sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import recbole_cdr_amd  # noqa: E402
from recbole_cdr_amd import binding as B_  # noqa: E402
from recbole_cdr_amd.data.synthetic import SyntheticCrossDomainDataset  # noqa: E402
from recbole_cdr_amd.model.cross_domain_recommender.conet import CoNet  # noqa: E402

dev = 'cuda:0'
ds = SyntheticCrossDomainDataset(OU=5983, TOU=20986, SOU=129127, OI=1, TOI=18563, SOI=115172, n_source_inter=400000, n_target_inter=200000)
cfg = {'source_domain': {'NEG_PREFIX': 'neg_'}, 'target_domain': {'NEG_PREFIX': 'neg_'}, 'device': dev, 'embedding_size': 128,
       'reg_weight': 0.01, 'mlp_hidden_size': [64, 32, 16, 8]}
torch.manual_seed(0)
model = CoNet(cfg, ds).to(dev)
rng = np.random.RandomState(0)
S, k = 819, 4
batches = [dict(ds.pointwise_batch('source', S, k, rng, dev), **ds.pointwise_batch('target', S, k, rng, dev)) for _ in range(4)]
if os.environ.get('MB_CLUSTER'):        # what-if: each domain's rows ordered by user id, i.e. the overlapped users' rows (id < n_overlap) in the first blocks
    for b in batches:
        for dom in ('source', 'target'):
            o = torch.argsort(b[f'{dom}_user_id'], stable=True)
            for f in ('user_id', 'item_id', 'label'):
                b[f'{dom}_{f}'] = b[f'{dom}_{f}'][o].contiguous()
    print('rows ordered by user id inside each domain (MB_CLUSTER)')
prof = None
if os.environ.get('CDR_CONET_PROF'):
    prof = torch.zeros(64 + 2 * 2048, dtype=torch.int64, device=dev)
    lib = B_.load()
    lib.cdr_conet_debug_prof.argtypes = [ctypes.c_void_p]
    assert lib.cdr_conet_debug_prof(ctypes.c_void_p(prof.data_ptr())) == 0
for i in range(5):
    model.zero_grad(set_to_none=True)
    model.calculate_loss(batches[i % 4]).backward()
torch.cuda.synchronize()
B_.timing_enable(dev, 4096)
for i in range(50):
    model.zero_grad(set_to_none=True)
    model.calculate_loss(batches[i % 4]).backward()
torch.cuda.synchronize()
acc = {}
for name, ms in B_.timing_collect(dev):
    acc.setdefault(name, []).append(ms)
R = 2 * S * (1 + k)
fl = 152.6e3 * R
for name, v in acc.items():
    v = np.array(v[5:])
    mult = {'conet_fb_kernel': 2, 'conet_fwd_kernel': 1, 'conet_bwd_kernel': 1, 'conet_wgrad_kernel': 1}.get(name, 0)
    print('%-22s avg %.1f us  min %.1f us  %s' % (name, v.mean() * 1e3, v.min() * 1e3,
                                                  ('%.1f TFLOP/s' % (fl * mult / (v.mean() * 1e-3) / 1e12)) if mult else ''))
if prof is not None:
    p = prof.cpu().numpy()
    if p[40]:        # conet_fb_kernel: entry, [gather, layers 0..3, output unit], [backward layers 3..0], tail start, end
        f = [p[40], p[43], p[44], p[45]] + list(p[:2 + 4 + 1 + 4]) + [p[41], p[42]]
        print('fwd+bwd stamps (us from kernel entry):', [round(float(x - f[0]) / 100.0, 2) for x in f])
        nb = (R + 31) // 32
        be = p[64:64 + 2 * nb].reshape(nb, 2).astype(np.float64) / 100.0        # per block: entry, exit (us)
        t0 = be[:, 0].min()
        print('blocks: entry %.2f .. %.2f us after the first one; exit %.2f .. %.2f us; block duration min %.2f median %.2f max %.2f us' % (
            be[:, 0].min() - t0, be[:, 0].max() - t0, be[:, 1].min() - t0, be[:, 1].max() - t0,
            (be[:, 1] - be[:, 0]).min(), np.median(be[:, 1] - be[:, 0]), (be[:, 1] - be[:, 0]).max()))
    else:
        f = p[:2 + 4 + 1]
        print('fwd stamps (us from start):', [round(float(x - f[0]) / 100.0, 2) for x in f])
        b = p[16:16 + 3 + 8]
        print('bwd stamps (us from start):', [round(float(x - b[0]) / 100.0, 2) for x in b])
