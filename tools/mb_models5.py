#!/usr/bin/env python3
"""Step time of the five models of SURVEY 8f-4 (and of SSCDR, SURVEY 8a row a8, which no BASELINE config names) at their default sizes (properties/model/*.yaml: D = 64, train_batch_size 2,048,
ml-1m -> ml-100k sized id space as BASELINE C1/C2): calculate_loss + backward + dense native Adam, eager and as one hipGraph
(graph_step.GraphedTrainStep), beside the oracle's torch-CPU restatement of the same step on the host (all cores / 1 thread).
Usage on an MI355X: python tools/mb_models5.py [--steps 200]"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))

from helpers import FakeDataset, base_config  # noqa: E402
from oracle.common import IdSpace  # noqa: E402

DEV = 'cuda:0'


def batch(ids, B, pairwise, rng):
    src_u = np.array(list(range(1, ids.OU)) + list(range(ids.OU + ids.TOU, ids.total_num_users)))
    src_i = np.array(list(range(1, ids.OI)) + list(range(ids.OI + ids.TOI, ids.total_num_items)))
    tgt_u, tgt_i = np.arange(1, ids.OU + ids.TOU), np.arange(1, ids.OI + ids.TOI)
    t = lambda a: torch.from_numpy(np.asarray(a, dtype=np.int64))
    out = {}
    for d, us, its in (('source', src_u, src_i), ('target', tgt_u, tgt_i)):
        out[f'{d}_user_id'], out[f'{d}_item_id'] = t(rng.choice(us, B)), t(rng.choice(its, B))
        if pairwise:
            out[f'neg_{d}_item_id'] = t(rng.choice(its, B))
        else:
            out[f'{d}_label'] = torch.from_numpy((rng.rand(B) < 0.5).astype(np.float32))
    return out


def timed(fn, steps, warmup=10):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


def cpu_timed(fn, budget=3.0):
    fn()
    n, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < budget:
        fn()
        n += 1
    return (time.perf_counter() - t0) / n * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--steps', type=int, default=200)
    ap.add_argument('--no-cpu', action='store_true')
    ap.add_argument('--only', default=None, help='run one case by name (for kernel traces)')
    a = ap.parse_args()
    from recbole_cdr_amd.graph_step import GraphedTrainStep
    from recbole_cdr_amd.trainer.trainer import DenseAdam
    from recbole_cdr_amd.model.cross_domain_recommender.clfm import CLFM
    from recbole_cdr_amd.model.cross_domain_recommender.dtcdr import DTCDR
    from recbole_cdr_amd.model.cross_domain_recommender.deepapf import DeepAPF
    from recbole_cdr_amd.model.cross_domain_recommender.natr import NATR
    from recbole_cdr_amd.model.cross_domain_recommender.dcdcsr import DCDCSR
    from oracle import clfm as o_clfm, dtcdr as o_dtcdr, deepapf as o_apf, natr as o_natr, dcdcsr as o_dc
    from recbole_cdr_amd.model.cross_domain_recommender.sscdr import SSCDR
    from oracle import sscdr as o_ss
    # item-overlap pair shaped like ml-1m -> ml-100k (OU = 1, 943 + 6,040 users, ~1,600 shared / 3,883 / 1,664 items)
    ids = IdSpace(OU=1, TOU=943, SOU=6040, OI=1603, TOI=61, SOI=2280)
    rng = np.random.RandomState(0)
    B = 2048
    n_inter = 100_000
    s_pairs = np.stack([rng.randint(ids.OU + ids.TOU, ids.total_num_users, n_inter),
                        rng.choice(np.r_[1:ids.OI, ids.OI + ids.TOI:ids.total_num_items], n_inter)], 1)
    t_pairs = np.stack([rng.randint(1, ids.OU + ids.TOU, n_inter), rng.randint(1, ids.OI + ids.TOI, n_inter)], 1)
    ds = FakeDataset(ids, s_pairs, t_pairs)
    ds.device = DEV
    rows = []
    cases = [
        ('CLFM', CLFM, dict(user_embedding_size=64, source_item_embedding_size=64, target_item_embedding_size=64,
                            share_embedding_size=32, alpha=0.5, reg_weight=1e-4), False, None,
         lambda P, b, m: o_clfm.calculate_loss(P, ids, b, 0.5, 1e-4)),
        ('DTCDR-NeuMF', DTCDR, dict(embedding_size=64, mlp_hidden_size=[32, 16], dropout_prob=0.0, base_model='NeuMF', alpha=0.5),
         False, None, lambda P, b, m: o_dtcdr.calculate_loss(P, ids, b, 0.5)),
        ('DeepAPF', DeepAPF, dict(embedding_size=64, beta=0.5), False, None, lambda P, b, m: o_apf.calculate_loss(P, ids, b)),
        ('NATR-phase2', NATR, dict(source_embedding_size=64, target_embedding_size=64, reg_weight=1e-3, max_inter_length=50),
         False, 'TARGET', None),
        ('DCDCSR-BPR', DCDCSR, dict(latent_factor_model='BPR', embedding_size=64, mlp_hidden_size=[128], k=10, map_batch_size=1024),
         True, 'TARGET', lambda P, b, m: o_dc.rec_loss(P, ids, b, 'TARGET')),
        # SSCDR (sscdr.py:89-195): the triplet-margin phase on squared-norm-normalised rows, and the map phase (MSE + lambda x triplet on
        # items drawn by the in-loss numpy sampler: host work inside the loss, so that phase is timed eagerly only)
        ('SSCDR-triplet', SSCDR, {'embedding_size': 64, 'margin': 0.2, 'mlp_hidden_size': [128], 'lambda': 0.1}, True, 'TARGET',
         lambda P, b, m: o_ss.calculate_loss(P, ids, b, 'TARGET', 0.2, 0.1)),
        ('SSCDR-map', SSCDR, {'embedding_size': 64, 'margin': 0.2, 'mlp_hidden_size': [128], 'lambda': 0.1}, True, 'OVERLAP', None),
        # ... and with the in-loss sampler on the device (config['sscdr_device_sampler']): no host work inside the loss, capturable
        ('SSCDR-map-device-sampler', SSCDR, {'embedding_size': 64, 'margin': 0.2, 'mlp_hidden_size': [128], 'lambda': 0.1, 'sscdr_device_sampler': True},
         True, 'OVERLAP', None),
    ]
    for name, cls, kw, pairwise, phase, oracle_loss in cases:
        if a.only and name != a.only:
            continue
        torch.manual_seed(0)
        model = cls(base_config(DEV, **kw), ds).to(DEV)
        if phase:
            model.set_phase(phase)
        model.train()
        opt = DenseAdam(model.parameters(), lr=1e-3)
        b = {k: v.to(DEV) for k, v in batch(ids, B, pairwise, rng).items()}
        if name.startswith('SSCDR-map'):
            b['overlap'] = torch.from_numpy(rng.choice(np.arange(1, ids.OI), 100, replace=False)).view(-1, 1).to(DEV)    # OB = 100 (default)

        def eager():
            opt.zero_grad(set_to_none=True)
            loss = model.calculate_loss(b)
            loss = loss.sum() if loss.dim() else loss
            loss.backward()
            opt.step()
        t_eager = timed(eager, a.steps)
        t_graph = None
        try:
            if name == 'SSCDR-map':
                raise RuntimeError('the numpy sampler runs on the host inside the loss (sscdr.py:94-111)')
            g = GraphedTrainStep(model, opt, b)
            t_graph = timed(lambda: g.graph.replay(), a.steps)
        except Exception as e:                                     # noqa: BLE001 -- report, do not hide
            t_graph = f'not capturable: {type(e).__name__}: {e}'
        row = {'model': name, 'rows_per_step': 100 if name.startswith('SSCDR-map') else B if name in ('NATR-phase2', 'DCDCSR-BPR', 'SSCDR-triplet') else 2 * B,
               'eager_ms': round(t_eager, 4), 'graph_ms': t_graph if isinstance(t_graph, str) else round(t_graph, 4)}
        if not a.no_cpu and (oracle_loss is not None or name == 'NATR-phase2'):
            P = {k: v.detach().cpu().clone().requires_grad_(v.requires_grad) for k, v in model.named_parameters()}
            bc = {k: v.cpu() for k, v in b.items()}
            if name == 'NATR-phase2':
                hist = o_natr.history_info(ids, t_pairs, 50)
                oracle_loss = lambda P, b, m: o_natr.calculate_loss(P, ids, hist, b, 'TARGET', 1e-3)   # noqa: E731
            copt = torch.optim.Adam([p for p in P.values() if p.requires_grad], lr=1e-3)

            def cpu_step():
                copt.zero_grad()
                oracle_loss(P, bc, None).sum().backward()
                copt.step()
            ncores = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
            try:                                         # the cgroup quota, not the 256 CPUs the container can see
                q_, p_ = open('/sys/fs/cgroup/cpu.max').read().split()
                if q_ != 'max':
                    ncores = min(ncores, max(1, -(-int(q_) // int(p_))))
            except Exception:
                pass
            for threads, key in ((ncores, 'cpu_all_cores_ms'), (1, 'cpu_one_thread_ms')):
                torch.set_num_threads(threads)
                row[key] = round(cpu_timed(cpu_step), 3)
            row['cpu_cores'] = ncores
        rows.append(row)
        print(json.dumps(row), flush=True)


if __name__ == '__main__':
    main()
