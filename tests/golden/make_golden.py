#!/usr/bin/env python3
"""Generate golden vectors by running the REFERENCE's own model code.

Run in the build container only (needs the read-only mount /root/reference):

    python -B tests/golden/make_golden.py

The reference's Python never enters this repository; only numbers do.  Each
``*.npz`` holds, for one (model, overlap mode, hyper-parameter) case:

    meta/*            scalar config (ints / floats / strings as 0-d arrays)
    param/<name>      every nn.Parameter after the reference's own init
    in/<field>        the interaction tensors handed to calculate_loss/predict
    loss/<phase>      calculate_loss output(s)
    grad/<phase>/<p>  d(sum of losses)/d(param) from the reference's autograd
    predict/<phase>   predict() output
    fullsort/<phase>  full_sort_predict() output

Third-party ``recbole`` is replaced by tests/golden/recbole_stub.py (see its
header for which symbols carry arithmetic and the resulting parity caveat).
Reference entry points exercised (file:line under /root/reference/recbole_cdr):
  model/cross_domain_recommender/emcdr.py:110-233, cmf.py:75-112,
  conet.py:105-242, sscdr.py:89-259, bitgcf.py:92-282,
  clfm.py:74-145, dtcdr.py:112-211, deepapf.py:69-175, natr.py:75-191, dcdcsr.py:90-280,
  data/dataset.py:151-249 (get_history_matrix, called unbound), :344-445, data/dataloader.py:114-162,240-247.
"""
import os
import sys

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import recbole_stub  # noqa: E402

recbole_stub.install()
sys.path.insert(1, '/root/reference')

import numpy as np  # noqa: E402
import pandas as pd  # noqa: E402
import scipy.sparse as sp  # noqa: E402
import torch  # noqa: E402

from recbole_cdr.model.cross_domain_recommender.emcdr import EMCDR  # noqa: E402
from recbole_cdr.model.cross_domain_recommender.cmf import CMF  # noqa: E402
from recbole_cdr.model.cross_domain_recommender.conet import CoNet  # noqa: E402
from recbole_cdr.model.cross_domain_recommender.sscdr import SSCDR  # noqa: E402
from recbole_cdr.model.cross_domain_recommender.bitgcf import BiTGCF  # noqa: E402
from recbole_cdr.model.cross_domain_recommender.clfm import CLFM  # noqa: E402
from recbole_cdr.model.cross_domain_recommender.dtcdr import DTCDR  # noqa: E402
from recbole_cdr.model.cross_domain_recommender.deepapf import DeepAPF  # noqa: E402
from recbole_cdr.model.cross_domain_recommender.natr import NATR  # noqa: E402
from recbole_cdr.model.cross_domain_recommender.dcdcsr import DCDCSR  # noqa: E402
from recbole_cdr.data.dataset import CrossDomainDataset, CrossDomainSingleDataset  # noqa: E402
from recbole_cdr.data.dataloader import CrossDomainDataloader, CrossDomainFullSortEvalDataLoader  # noqa: E402
from recbole_cdr.utils import CrossDomainDataLoaderState  # noqa: E402

torch.set_num_threads(1)


# ----------------------------------------------------------------------------------------------
# duck-typed dataset
class _Single:
    def __init__(self, domain, n_user, n_item, inter_feat=None):
        self.uid_field = f'{domain}_user_id'
        self.iid_field = f'{domain}_item_id'
        self.label_field = f'{domain}_label'
        self._n = {self.uid_field: n_user, self.iid_field: n_item}
        self.inter_feat = inter_feat

    def num(self, field):
        return self._n[field]


class _Dataset:
    """ids: [0]=PAD, [1,OU) overlap, [OU,OU+TO) target-only, [OU+TO,total) source-only (dataset.py:384-399)."""

    def __init__(self, OU, TOU, SOU, OI, TOI, SOI, seed):
        self.OU, self.TOU, self.SOU, self.OI, self.TOI, self.SOI = OU, TOU, SOU, OI, TOI, SOI
        self.num_overlap_user, self.num_overlap_item = OU, OI
        self.num_target_only_user, self.num_source_only_user = TOU, SOU
        self.num_target_only_item, self.num_source_only_item = TOI, SOI
        self.num_total_user = OU + TOU + SOU
        self.num_total_item = OI + TOI + SOI
        self.overlap_id_field = 'overlap'
        rng = np.random.RandomState(seed)
        self.src_users = np.array(list(range(1, OU)) + list(range(OU + TOU, self.num_total_user)))
        self.src_items = np.array(list(range(1, OI)) + list(range(OI + TOI, self.num_total_item)))
        self.tgt_users = np.arange(1, OU + TOU)
        self.tgt_items = np.arange(1, OI + TOI)
        n_s, n_t = 160, 140
        su, si = rng.choice(self.src_users, n_s), rng.choice(self.src_items, n_s)
        tu, ti = rng.choice(self.tgt_users, n_t), rng.choice(self.tgt_items, n_t)
        s_pairs = np.unique(np.stack([su, si], 1), axis=0)
        t_pairs = np.unique(np.stack([tu, ti], 1), axis=0)
        self.s_pairs, self.t_pairs = s_pairs, t_pairs
        s_feat = {'source_user_id': torch.from_numpy(s_pairs[:, 0].copy()), 'source_item_id': torch.from_numpy(s_pairs[:, 1].copy())}
        t_feat = {'target_user_id': torch.from_numpy(t_pairs[:, 0].copy()), 'target_item_id': torch.from_numpy(t_pairs[:, 1].copy())}
        self.source_domain_dataset = _Single('source', OU + SOU, OI + SOI, s_feat)
        self.target_domain_dataset = _Single('target', OU + TOU, OI + TOI, t_feat)

    def inter_matrix(self, form='coo', value_field=None, domain='source'):
        p = self.s_pairs if domain == 'source' else self.t_pairs
        m = sp.coo_matrix((np.ones(len(p), dtype=np.float32), (p[:, 0], p[:, 1])),
                          shape=(self.num_total_user, self.num_total_item))
        return m

    def meta(self):
        return dict(OU=self.OU, TOU=self.TOU, SOU=self.SOU, OI=self.OI, TOI=self.TOI, SOI=self.SOI)

    # -- history matrices: the REFERENCE's get_history_matrix (dataset.py:200-249) run unbound on a duck-typed single dataset,
    #    sized by the union id space as CrossDomainDataset.history_{user,item}_matrix do (dataset.py:596-654)
    def _history(self, domain, row):
        class _Feat(dict):
            def __len__(self):
                return len(next(iter(self.values())))

        class _S:
            pass
        d = self.source_domain_dataset if domain == 'source' else self.target_domain_dataset
        s = _S()
        s.uid_field, s.iid_field = d.uid_field, d.iid_field
        s.inter_feat = _Feat(d.inter_feat)
        s._check_field = lambda *a: None
        s.logger = type('L', (), {'warning': staticmethod(lambda *a, **k: None)})()
        return CrossDomainSingleDataset.get_history_matrix(s, self.num_total_user, self.num_total_item, row=row)

    def history_item_matrix(self, value_field=None, domain='source'):
        return self._history(domain, 'user')

    def history_user_matrix(self, value_field=None, domain='source'):
        return self._history(domain, 'item')


def users_overlap_ds(seed=1):
    return _Dataset(OU=12, TOU=10, SOU=14, OI=1, TOI=20, SOI=24, seed=seed)


def items_overlap_ds(seed=2):
    return _Dataset(OU=1, TOU=15, SOU=18, OI=10, TOI=12, SOI=13, seed=seed)


def base_config(**kw):
    cfg = {'source_domain': {'NEG_PREFIX': 'neg_'}, 'target_domain': {'NEG_PREFIX': 'neg_'}, 'device': 'cpu'}
    cfg.update(kw)
    return cfg


# ----------------------------------------------------------------------------------------------
# batches in recbole's layout (k-major negatives; POINTWISE = repeat(1+k) + labels)
def pairwise_batch(ds, rng, S, k, prefix, force_ids=True):
    if prefix == 'source':
        users, items = ds.src_users, ds.src_items
    else:
        users, items = ds.tgt_users, ds.tgt_items
    u = rng.choice(users, S)
    i = rng.choice(items, S)
    if force_ids and S >= 4:
        u[0] = users[0]; u[1] = u[2]            # a duplicate user inside the batch
        i[3] = i[0]                              # a duplicate positive item
    neg = rng.choice(items, S * k)
    return {f'{prefix}_user_id': torch.from_numpy(np.tile(u, k)),
            f'{prefix}_item_id': torch.from_numpy(np.tile(i, k)),
            f'neg_{prefix}_item_id': torch.from_numpy(neg)}


def pointwise_batch(ds, rng, S, k, prefix, with_pad=True):
    if prefix == 'source':
        users, items = ds.src_users, ds.src_items
    else:
        users, items = ds.tgt_users, ds.tgt_items
    u = rng.choice(users, S)
    i = rng.choice(items, S)
    if with_pad and S >= 4:
        u[0] = 0; i[1] = 0                       # PAD id 0 is a real row
        u[2] = u[3]
    neg = rng.choice(items, S * k)
    uu = np.tile(u, 1 + k)
    ii = np.concatenate([i, neg])
    lab = np.concatenate([np.ones(S), np.zeros(S * k)]).astype(np.float32)
    return {f'{prefix}_user_id': torch.from_numpy(uu), f'{prefix}_item_id': torch.from_numpy(ii),
            f'{prefix}_label': torch.from_numpy(lab)}


def overlap_batch(n_overlap, rng, OB):
    idx = rng.permutation(n_overlap)[:OB].astype(np.int64)
    return {'overlap': torch.from_numpy(idx.reshape(-1, 1))}   # [OB,1]  


# ----------------------------------------------------------------------------------------------
def _np(t):
    return t.detach().cpu().numpy().copy()


def loss_and_grads(model, interaction, out, phase):
    model.zero_grad(set_to_none=True)
    losses = model.calculate_loss(interaction)
    if isinstance(losses, tuple):
        total = sum(losses)
        out[f'loss/{phase}'] = np.array([float(l) if l.dim() == 0 else float(l.reshape(-1)[0]) for l in losses], dtype=np.float32)
    else:
        total = losses
        out[f'loss/{phase}'] = _np(losses).reshape(-1).astype(np.float32)
    total = total.sum()
    total.backward()
    for name, p in model.named_parameters():
        if p.grad is not None:
            out[f'grad/{phase}/{name}'] = _np(p.grad)


def dump(name, out):
    path = os.path.join(HERE, name + '.npz')
    arrays = {}
    for k, v in out.items():
        a = np.asarray(v)
        arrays[k] = a
    np.savez_compressed(path, **arrays)
    print(f'{name}.npz  {os.path.getsize(path) / 1024:.1f} KiB  ({len(arrays)} arrays)')


def put_meta(out, ds, **kw):
    for k, v in ds.meta().items():
        out[f'meta/{k}'] = np.int64(v)
    for k, v in kw.items():
        out[f'meta/{k}'] = np.asarray(v)


def put_params(out, model):
    for name, p in model.named_parameters():
        out[f'param/{name}'] = _np(p)


def put_inputs(out, inter):
    for k, v in inter.items():
        out[f'in/{k}'] = _np(v)


# ----------------------------------------------------------------------------------------------
def gen_emcdr():
    for mode, mk in (('users', users_overlap_ds), ('items', items_overlap_ds)):
        for lfm, k in (('MF', 1), ('BPR', 1), ('BPR', 4)):
            for mapf in ('linear', 'non_linear'):
                D = 8 if mapf == 'linear' else 16
                torch.manual_seed(100 + k + len(mapf) + len(mode))
                ds = mk()
                cfg = base_config(latent_factor_model=lfm, source_embedding_size=D, target_embedding_size=D,
                                  reg_weight=0.01, mapping_function=mapf, mlp_hidden_size=[12])
                model = EMCDR(cfg, ds)
                rng = np.random.RandomState(7 * k + D)
                S = 12
                inter = {}
                if lfm == 'MF':
                    inter.update(pointwise_batch(ds, rng, S, k, 'source', with_pad=False))
                    inter.update(pointwise_batch(ds, rng, S + 3, k, 'target', with_pad=False))   # unequal lengths (Q11)
                else:
                    inter.update(pairwise_batch(ds, rng, S, k, 'source'))
                    inter.update(pairwise_batch(ds, rng, S + 3, k, 'target'))
                n_ov = ds.OU if mode == 'users' else ds.OI
                inter.update(overlap_batch(n_ov, rng, OB=7))
                out = {}
                put_meta(out, ds, model='EMCDR', mode=mode, latent_factor_model=lfm, k=k, mapping_function=mapf,
                         D=D, reg_weight=0.01, mlp_hidden_size=np.array([12]), S=S)
                put_params(out, model)
                put_inputs(out, inter)
                for phase in ('SOURCE', 'TARGET', 'OVERLAP', 'BOTH'):
                    model.set_phase(phase)
                    loss_and_grads(model, inter, out, phase)
                # eval users: overlapped, PAD, target-only
                eu = np.array([0, 1, ds.OU + 1 if ds.TOU else 1, ds.OU + ds.TOU - 1, 2 if ds.OU > 2 else 1], dtype=np.int64)
                ei_t = rng.choice(np.arange(0, ds.OI + ds.TOI), len(eu)).astype(np.int64)
                su = rng.choice(ds.src_users, 5).astype(np.int64)
                si = rng.choice(ds.src_items, 5).astype(np.int64)
                ev = {'target_user_id': torch.from_numpy(eu), 'target_item_id': torch.from_numpy(ei_t),
                      'source_user_id': torch.from_numpy(su), 'source_item_id': torch.from_numpy(si)}
                for kk, v in ev.items():
                    out[f'evalin/{kk}'] = _np(v)
                with torch.no_grad():
                    for phase in ('SOURCE', 'TARGET', 'OVERLAP', 'BOTH'):
                        model.set_phase(phase)
                        out[f'predict/{phase}'] = _np(model.predict(ev))
                        out[f'fullsort/{phase}'] = _np(model.full_sort_predict(ev))
                dump(f'emcdr_{mode}_{lfm.lower()}{k}_{mapf}', out)


def gen_cmf():
    for mode, mk in (('users', users_overlap_ds), ('items', items_overlap_ds)):
        torch.manual_seed(31 + len(mode))
        ds = mk()
        cfg = base_config(embedding_size=16, alpha=0.3, **{'lambda': 0.02, 'gamma': 0.05})
        model = CMF(cfg, ds)
        rng = np.random.RandomState(5)
        inter = {}
        inter.update(pointwise_batch(ds, rng, 10, 1, 'source'))
        inter.update(pointwise_batch(ds, rng, 9, 2, 'target'))
        out = {}
        put_meta(out, ds, model='CMF', mode=mode, D=16, alpha=0.3, lam=0.02, gamma=0.05)
        put_params(out, model)
        put_inputs(out, inter)
        loss_and_grads(model, inter, out, 'BOTH')
        eu = np.array([0, 1, 3, ds.OU + ds.TOU - 1], dtype=np.int64)
        ei = rng.choice(np.arange(0, ds.OI + ds.TOI), len(eu)).astype(np.int64)
        ev = {'target_user_id': torch.from_numpy(eu), 'target_item_id': torch.from_numpy(ei)}
        for kk, v in ev.items():
            out[f'evalin/{kk}'] = _np(v)
        with torch.no_grad():
            out['predict/BOTH'] = _np(model.predict(ev))
            out['fullsort/BOTH'] = _np(model.full_sort_predict(ev))
        dump(f'cmf_{mode}', out)


def gen_conet():
    for mode, mk in (('users', users_overlap_ds), ('items', items_overlap_ds)):
        for tag, D, layers in (('a', 8, [12, 8, 4]), ('b', 16, [64, 32, 16, 8])):
            torch.manual_seed(77 + D)
            ds = mk()
            cfg = base_config(embedding_size=D, reg_weight=0.01, mlp_hidden_size=layers)
            model = CoNet(cfg, ds)
            # give the biases non-zero values so the bias path is exercised (init sets them to 0)
            with torch.no_grad():
                for n, p in model.named_parameters():
                    if n.endswith('bias'):
                        p.normal_(0, 0.1)
            rng = np.random.RandomState(11 + D)
            inter = {}
            inter.update(pointwise_batch(ds, rng, 9, 4, 'source'))    # 1 pos + 4 labelled negatives (BASELINE C3)
            inter.update(pointwise_batch(ds, rng, 7, 4, 'target'))
            out = {}
            put_meta(out, ds, model='CoNet', mode=mode, D=D, mlp_hidden_size=np.array(layers))
            put_params(out, model)
            put_inputs(out, inter)
            loss_and_grads(model, inter, out, 'BOTH')
            with torch.no_grad():
                out['fwd/source'] = _np(model.source_forward(inter['source_user_id'], inter['source_item_id']))
                out['fwd/target'] = _np(model.target_forward(inter['target_user_id'], inter['target_item_id']))
            eu = np.array([0, 1, 3, ds.OU + ds.TOU - 1], dtype=np.int64)
            ei = rng.choice(np.arange(0, ds.OI + ds.TOI), len(eu)).astype(np.int64)
            ev = {'target_user_id': torch.from_numpy(eu), 'target_item_id': torch.from_numpy(ei)}
            for kk, v in ev.items():
                out[f'evalin/{kk}'] = _np(v)
            with torch.no_grad():
                out['predict/BOTH'] = _np(model.predict(ev))            # [B,1]
                out['fullsort/BOTH'] = _np(model.full_sort_predict(ev))  # [U,N]
            dump(f'conet_{mode}_{tag}', out)


def gen_sscdr():
    for mode, mk in (('users', users_overlap_ds), ('items', items_overlap_ds)):
        torch.manual_seed(55 + len(mode))
        ds = mk()
        D = 16
        cfg = base_config(embedding_size=D, margin=0.7, mlp_hidden_size=[12], **{'lambda': 0.25})
        model = SSCDR(cfg, ds)
        with torch.no_grad():
            # scale some rows so that ||e||^2 > 1 : both branches of embedding_normalize (sscdr.py:120-124)
            for emb in (model.source_user_embedding, model.source_item_embedding,
                        model.target_user_embedding, model.target_item_embedding):
                emb.weight[::3] *= 6.0
            for n, p in model.named_parameters():
                if n.endswith('bias'):
                    p.normal_(0, 0.1)
        rng = np.random.RandomState(23)
        inter = {}
        inter.update(pairwise_batch(ds, rng, 10, 2, 'source'))
        inter.update(pairwise_batch(ds, rng, 11, 2, 'target'))
        n_ov = ds.OU if mode == 'users' else ds.OI
        inter.update(overlap_batch(n_ov, rng, OB=6))
        out = {}
        put_meta(out, ds, model='SSCDR', mode=mode, D=D, margin=0.7, lam=0.25, mlp_hidden_size=np.array([12]))
        put_params(out, model)
        put_inputs(out, inter)
        # interaction lists the model built (sscdr.py:74-87) in CSR form + sampled ids for the overlap batch
        lists = model.user_interacted_items if mode == 'users' else model.item_interacted_users
        out['aux/hist_indptr'] = np.cumsum([0] + [len(x) for x in lists]).astype(np.int64)
        out['aux/hist_indices'] = np.array([y for x in lists for y in x], dtype=np.int64)
        idx = inter['overlap'].squeeze(1)
        np.random.seed(99)
        pos, neg = model.sample(idx, mode='user' if mode == 'users' else 'item')
        out['aux/sampled_pos'] = _np(pos)
        out['aux/sampled_neg'] = _np(neg)
        for phase in ('SOURCE', 'TARGET', 'BOTH'):
            model.set_phase(phase)
            loss_and_grads(model, inter, out, phase)
        model.set_phase('OVERLAP')
        np.random.seed(99)      # same draws as recorded above
        loss_and_grads(model, inter, out, 'OVERLAP')
        eu = np.array([0, 1, ds.OU + 1 if ds.TOU else 1, ds.OU + ds.TOU - 1], dtype=np.int64)
        ei_t = rng.choice(np.arange(0, ds.OI + ds.TOI), len(eu)).astype(np.int64)
        su = rng.choice(ds.src_users, 4).astype(np.int64)
        si = rng.choice(ds.src_items, 4).astype(np.int64)
        ev = {'target_user_id': torch.from_numpy(eu), 'target_item_id': torch.from_numpy(ei_t),
              'source_user_id': torch.from_numpy(su), 'source_item_id': torch.from_numpy(si)}
        for kk, v in ev.items():
            out[f'evalin/{kk}'] = _np(v)
        with torch.no_grad():
            for phase in ('SOURCE', 'TARGET', 'OVERLAP'):
                model.set_phase(phase)
                out[f'predict/{phase}'] = _np(model.predict(ev))
                out[f'fullsort/{phase}'] = _np(model.full_sort_predict(ev))
        dump(f'sscdr_{mode}', out)


def gen_bitgcf():
    # scipy >= 1.8 dropped dok_matrix._update; the reference calls it (bitgcf.py:101). Harness-side shim.
    if not hasattr(sp.dok_matrix, '_update'):
        sp.dok_matrix._update = lambda self, d: [self.__setitem__(k, v) for k, v in d.items()]
    for mode, mk in (('users', users_overlap_ds), ('items', items_overlap_ds)):
        for connect in ('concat', 'mean'):
            torch.manual_seed(91 + len(connect))
            ds = mk()
            D = 8
            cfg = base_config(embedding_size=D, n_layers=2, reg_weight=0.001, lambda_source=0.8, lambda_target=0.7,
                              drop_rate=0.0, connect_way=connect)
            model = BiTGCF(cfg, ds)
            rng = np.random.RandomState(3)
            inter = {}
            inter.update(pointwise_batch(ds, rng, 10, 1, 'source'))
            inter.update(pointwise_batch(ds, rng, 8, 1, 'target'))
            out = {}
            put_meta(out, ds, model='BiTGCF', mode=mode, D=D, n_layers=2, reg_weight=0.001, lambda_source=0.8,
                     lambda_target=0.7, connect_way=connect)
            put_params(out, model)
            put_inputs(out, inter)
            out['aux/s_pairs'] = ds.s_pairs.astype(np.int64)
            out['aux/t_pairs'] = ds.t_pairs.astype(np.int64)
            for dom, adj in (('source', model.source_norm_adj_matrix), ('target', model.target_norm_adj_matrix)):
                a = adj.coalesce()
                out[f'aux/adj_{dom}_idx'] = _np(a.indices())
                out[f'aux/adj_{dom}_val'] = _np(a.values())
            model.train()
            loss_and_grads(model, inter, out, 'BOTH')
            with torch.no_grad():
                su, si, tu, ti = model.forward()
                out['fwd/source_user'] = _np(su); out['fwd/source_item'] = _np(si)
                out['fwd/target_user'] = _np(tu); out['fwd/target_item'] = _np(ti)
            eu = np.array([0, 1, 3, ds.OU + ds.TOU - 1], dtype=np.int64)
            ei = rng.choice(np.arange(0, ds.OI + ds.TOI), len(eu)).astype(np.int64)
            ev = {'target_user_id': torch.from_numpy(eu), 'target_item_id': torch.from_numpy(ei)}
            for kk, v in ev.items():
                out[f'evalin/{kk}'] = _np(v)
            model.eval()
            with torch.no_grad():
                out['predict/BOTH'] = _np(model.predict(ev))
                out['fullsort/BOTH'] = _np(model.full_sort_predict(ev))
            dump(f'bitgcf_{mode}_{connect}', out)


def _eval_inputs(ds, rng, out, n=4, source=False):
    eu = np.array([0, 1, 3, ds.OU + ds.TOU - 1], dtype=np.int64)[:n]
    ei = rng.choice(np.arange(0, ds.OI + ds.TOI), len(eu)).astype(np.int64)
    ev = {'target_user_id': torch.from_numpy(eu), 'target_item_id': torch.from_numpy(ei)}
    if source:
        ev['source_user_id'] = torch.from_numpy(rng.choice(ds.src_users, len(eu)).astype(np.int64))
        ev['source_item_id'] = torch.from_numpy(rng.choice(ds.src_items, len(eu)).astype(np.int64))
    for kk, v in ev.items():
        out[f'evalin/{kk}'] = _np(v)
    return ev


def _randomise_biases(model):
    with torch.no_grad():
        for n, p in model.named_parameters():
            if n.endswith('bias'):
                p.normal_(0, 0.1)


def gen_clfm():
    """clfm.py:74-145.  ``target_item_embedding_size`` is read from ``source_item_embedding_size`` (clfm.py:39)."""
    for tag, mk, share in (('users', users_overlap_ds, 6), ('items', items_overlap_ds, 6), ('users_allshared', users_overlap_ds, 16)):
        torch.manual_seed(41 + share + len(tag))
        ds = mk()
        cfg = base_config(user_embedding_size=12, source_item_embedding_size=16, target_item_embedding_size=16,
                          share_embedding_size=share, alpha=0.3, reg_weight=0.01)
        model = CLFM(cfg, ds)
        rng = np.random.RandomState(17 + share)
        inter = {}
        inter.update(pointwise_batch(ds, rng, 10, 1, 'source'))
        inter.update(pointwise_batch(ds, rng, 9, 2, 'target'))
        out = {}
        put_meta(out, ds, model='CLFM', mode=tag.split('_')[0], user_embedding_size=12, item_embedding_size=16,
                 share_embedding_size=share, alpha=0.3, reg_weight=0.01)
        put_params(out, model)
        put_inputs(out, inter)
        loss_and_grads(model, inter, out, 'BOTH')
        ev = _eval_inputs(ds, rng, out)
        with torch.no_grad():
            out['predict/BOTH'] = _np(model.predict(ev))
            out['fullsort/BOTH'] = _np(model.full_sort_predict(ev))
        dump(f'clfm_{tag}', out)


def gen_dtcdr():
    """dtcdr.py:112-126 (neumf_forward), :182-199 (calculate_loss), :201-207 (predict); base_model = NeuMF, dropout_prob = 0
    (the dropout mask is a torch-RNG stream, not comparable across implementations)."""
    if not hasattr(np, 'NINF'):          # numpy >= 2 dropped the alias the reference uses (dtcdr.py:55-59). Harness-side shim.
        np.NINF = -np.inf
    for mode, mk in (('users', users_overlap_ds), ('items', items_overlap_ds)):
        torch.manual_seed(61 + len(mode))
        ds = mk()
        cfg = base_config(embedding_size=8, mlp_hidden_size=[12, 6], dropout_prob=0.0, base_model='NeuMF', alpha=0.4)
        model = DTCDR(cfg, ds)
        _randomise_biases(model)
        with torch.no_grad():                          # exact ties of torch.maximum (gradient split 1/2 : 1/2)
            model.target_user_embedding.weight[1, :3] = model.source_user_embedding.weight[1, :3]
            model.target_item_embedding.weight[1, 2:5] = model.source_item_embedding.weight[1, 2:5]
        rng = np.random.RandomState(29)
        inter = {}
        inter.update(pointwise_batch(ds, rng, 10, 1, 'source'))
        inter.update(pointwise_batch(ds, rng, 9, 2, 'target'))
        inter['source_user_id'][4] = 1; inter['source_item_id'][5] = 1 if ds.OI > 1 else inter['source_item_id'][5]
        inter['target_user_id'][4] = 1; inter['target_item_id'][5] = 1
        out = {}
        put_meta(out, ds, model='DTCDR', mode=mode, D=8, mlp_hidden_size=np.array([12, 6]), alpha=0.4, base_model='NeuMF')
        put_params(out, model)
        put_inputs(out, inter)
        model.train()
        loss_and_grads(model, inter, out, 'BOTH')
        ev = _eval_inputs(ds, rng, out)
        model.eval()
        with torch.no_grad():
            out['predict/BOTH'] = _np(model.predict(ev))
        dump(f'dtcdr_{mode}_neumf', out)


def gen_deepapf():
    """deepapf.py:69-152 (source/target_forward, both modes), :160-175 (calculate_loss), :154-158 (predict)."""
    for mode, mk in (('users', users_overlap_ds), ('items', items_overlap_ds)):
        torch.manual_seed(83 + len(mode))
        ds = mk()
        cfg = base_config(embedding_size=8, beta=0.5)
        model = DeepAPF(cfg, ds)
        _randomise_biases(model)
        rng = np.random.RandomState(31)
        inter = {}
        inter.update(pointwise_batch(ds, rng, 10, 1, 'source'))
        inter.update(pointwise_batch(ds, rng, 9, 2, 'target'))
        n_ov = ds.OU if mode == 'users' else ds.OI
        key = 'user_id' if mode == 'users' else 'item_id'
        inter[f'source_{key}'][5] = n_ov if mode == 'items' else inter[f'source_{key}'][5]   # id == overlapped_num: NOT masked (">" in deepapf.py:75)
        inter[f'target_{key}'][5] = n_ov
        inter[f'target_{key}'][6] = n_ov - 1
        out = {}
        put_meta(out, ds, model='DeepAPF', mode=mode, D=8)
        put_params(out, model)
        put_inputs(out, inter)
        loss_and_grads(model, inter, out, 'BOTH')
        with torch.no_grad():
            out['fwd/source'] = _np(model.source_forward(inter['source_user_id'], inter['source_item_id']))
            out['fwd/target'] = _np(model.target_forward(inter['target_user_id'], inter['target_item_id']))
        ev = _eval_inputs(ds, rng, out)
        with torch.no_grad():
            out['predict/BOTH'] = _np(model.predict(ev))
        dump(f'deepapf_{mode}', out)


def gen_natr():
    """natr.py:75-96 (history info), :98-110 (phase 1), :112-156 (phase 2, both modes), :158-175, :177-191 (predict)."""
    for mode, mk in (('users', users_overlap_ds), ('items', items_overlap_ds)):
        torch.manual_seed(97 + len(mode))
        ds = mk()
        cfg = base_config(source_embedding_size=8, target_embedding_size=12, reg_weight=1e-3, max_inter_length=5)
        model = NATR(cfg, ds)
        _randomise_biases(model)
        rng = np.random.RandomState(37)
        inter = {}
        inter.update(pointwise_batch(ds, rng, 10, 1, 'source'))
        inter.update(pointwise_batch(ds, rng, 9, 2, 'target'))
        out = {}
        put_meta(out, ds, model='NATR', mode=mode, Ds=8, Dt=12, reg_weight=1e-3, max_inter_length=5)
        put_params(out, model)
        put_inputs(out, inter)
        out['aux/s_pairs'] = ds.s_pairs.astype(np.int64)
        out['aux/t_pairs'] = ds.t_pairs.astype(np.int64)
        hist = model.history_user_matrix if mode == 'users' else model.history_item_matrix
        out['aux/history_matrix'] = _np(hist)
        out['aux/history_lens'] = _np(model.history_lens)
        out['aux/mask_mat'] = _np(model.mask_mat)
        ev = _eval_inputs(ds, rng, out, source=True)
        for phase in ('SOURCE', 'TARGET'):            # TARGET freezes the source tables (natr.py:69-73): keep this order
            model.set_phase(phase)
            loss_and_grads(model, inter, out, phase)
            with torch.no_grad():
                out[f'predict/{phase}'] = _np(model.predict(ev))
        dump(f'natr_{mode}', out)


class _DenseDataset(_Dataset):
    """Every id of either domain interacts at least once (DCDCSR divides by popularity sums: dcdcsr.py:117-133)."""

    def __init__(self, *a, **k):
        super().__init__(*a, **k)
        rng = np.random.RandomState(1234)
        s_extra = [np.stack([self.src_users, rng.choice(self.src_items, len(self.src_users))], 1),
                   np.stack([rng.choice(self.src_users, len(self.src_items)), self.src_items], 1)]
        t_extra = [np.stack([self.tgt_users, rng.choice(self.tgt_items, len(self.tgt_users))], 1),
                   np.stack([rng.choice(self.tgt_users, len(self.tgt_items)), self.tgt_items], 1)]
        self.s_pairs = np.unique(np.concatenate([self.s_pairs] + s_extra), axis=0)
        self.t_pairs = np.unique(np.concatenate([self.t_pairs] + t_extra), axis=0)
        rng.shuffle(self.s_pairs); rng.shuffle(self.t_pairs)          # history order = order of appearance
        self.source_domain_dataset.inter_feat = {'source_user_id': torch.from_numpy(self.s_pairs[:, 0].copy()),
                                                 'source_item_id': torch.from_numpy(self.s_pairs[:, 1].copy())}
        self.target_domain_dataset.inter_feat = {'target_user_id': torch.from_numpy(self.t_pairs[:, 0].copy()),
                                                 'target_item_id': torch.from_numpy(self.t_pairs[:, 1].copy())}


def gen_dcdcsr():
    """dcdcsr.py:90-110 (set_phase), :112-127 (BPR), :129-165 (benchmark), :168-192 (map loss), :194-280."""
    for mode, args in (('users', dict(OU=12, TOU=10, SOU=14, OI=1, TOI=20, SOI=24, seed=1)),
                       ('items', dict(OU=1, TOU=15, SOU=18, OI=10, TOI=12, SOI=13, seed=2))):
        torch.manual_seed(113 + len(mode))
        ds = _DenseDataset(**args)
        cfg = base_config(latent_factor_model='BPR', embedding_size=8, mlp_hidden_size=[12], k=3, map_batch_size=9)
        model = DCDCSR(cfg, ds)
        _randomise_biases(model)
        rng = np.random.RandomState(41)
        inter = {}
        inter.update(pairwise_batch(ds, rng, 10, 2, 'source'))
        inter.update(pairwise_batch(ds, rng, 11, 2, 'target'))
        out = {}
        put_meta(out, ds, model='DCDCSR', mode=mode, D=8, mlp_hidden_size=np.array([12]), k=3, map_batch_size=9)
        put_params(out, model)
        put_inputs(out, inter)
        out['aux/s_pairs'] = ds.s_pairs.astype(np.int64)
        out['aux/t_pairs'] = ds.t_pairs.astype(np.int64)
        unit = 'user' if mode == 'users' else 'item'
        out['aux/source_pop'] = _np(getattr(model, f'source_{unit}2pop'))
        out['aux/target_pop'] = _np(getattr(model, f'target_{unit}2pop'))
        ev = _eval_inputs(ds, rng, out, source=True)

        def evals(tag):
            with torch.no_grad():
                out[f'predict/{tag}'] = _np(model.predict(ev))
                out[f'fullsort/{tag}'] = _np(model.full_sort_predict(ev))
        model.set_phase('SOURCE'); loss_and_grads(model, inter, out, 'SOURCE'); evals('SOURCE')
        model.set_phase('TARGET'); loss_and_grads(model, inter, out, 'TARGET'); evals('TARGET')
        model.set_phase('BOTH')
        out['fwd/benchmark_embedding'] = _np(model.benchmark_embedding)
        n_units = ds.OU + ds.TOU if mode == 'users' else ds.OI + ds.TOI
        np.random.seed(77)
        out['aux/sampled_index'] = np.random.randint(0, n_units, 9).astype(np.int64)
        np.random.seed(77)
        loss_and_grads(model, inter, out, 'BOTH')
        model.set_phase('TARGET')
        out['fwd/affine_embedding'] = _np(model.affine_embedding)
        loss_and_grads(model, inter, out, 'TARGET2'); evals('TARGET2')
        dump(f'dcdcsr_{mode}', out)


# ----------------------------------------------------------------------------------------------
def gen_remap():
    """dataset.py:344-445 on raw string tokens, called unbound on a duck-typed object."""
    class _S:
        pass

    def run(case, s_users, s_items, t_users, t_items, s_user_feat=None, t_user_feat=None):
        obj = _S()
        obj.source_domain_dataset = _S(); obj.target_domain_dataset = _S()
        for d, us, its, uf in ((obj.source_domain_dataset, s_users, s_items, s_user_feat),
                               (obj.target_domain_dataset, t_users, t_items, t_user_feat)):
            d.uid_field, d.iid_field = 'user_id', 'item_id'
            d.inter_feat = pd.DataFrame({'user_id': us, 'item_id': its})
            d.user_feat = None if uf is None else pd.DataFrame({'user_id': uf})
            d.item_feat = None
        su, si, tu, ti = CrossDomainDataset.calculate_user_item_from_both_domain(obj)
        out = {}

        def put(prefix, cm):
            d = dict(cm)
            keys = sorted(d.keys())
            out[f'{prefix}/tokens'] = np.array(keys, dtype=object).astype('U')
            out[f'{prefix}/ids'] = np.array([d[k] for k in keys], dtype=np.int64)
        put('source_user', su); put('source_item', si); put('target_user', tu); put('target_item', ti)
        for k in ('num_overlap_user', 'num_source_only_user', 'num_target_only_user', 'num_total_user',
                  'num_overlap_item', 'num_source_only_item', 'num_target_only_item', 'num_total_item'):
            out[f'count/{k}'] = np.int64(getattr(obj, k))
        for nm, col in (('source_user', s_users), ('source_item', s_items), ('target_user', t_users), ('target_item', t_items)):
            out[f'in/{nm}_isnan'] = np.array([isinstance(x, float) and x != x for x in col], dtype=bool)
        out['in/source_user_tokens'] = np.array([str(x) for x in s_users]).astype('U')
        out['in/source_item_tokens'] = np.array([str(x) for x in s_items]).astype('U')
        out['in/target_user_tokens'] = np.array([str(x) for x in t_users]).astype('U')
        out['in/target_item_tokens'] = np.array([str(x) for x in t_items]).astype('U')
        if s_user_feat is not None:
            out['in/source_user_feat_tokens'] = np.array([str(x) for x in s_user_feat]).astype('U')
        if t_user_feat is not None:
            out['in/target_user_feat_tokens'] = np.array([str(x) for x in t_user_feat]).astype('U')
        # applied remap (dataset.py:109-123: inter_feat[field].map(lambda x: d.get(x, x)))
        def applied(cm, col):      # NaN tokens stay NaN in the reference (d.get(x, x)); recorded as -1 here
            d = dict(cm)
            return np.array([-1 if (isinstance(x, float) and x != x) else d.get(x, x) for x in col], dtype=np.int64)
        out['applied/source_user'] = applied(su, s_users)
        out['applied/source_item'] = applied(si, s_items)
        out['applied/target_user'] = applied(tu, t_users)
        out['applied/target_item'] = applied(ti, t_items)
        dump(f'remap_{case}', out)

    # lexicographic order: 'u10' < 'u2'; user overlap only
    run('lex_users',
        ['u2', 'u10', 'u1', 'u33', 'u2', 'a9', 'Z1'], ['sA', 'sB', 'sA', 'sC', 'sD', 'sB', 'sE'],
        ['u10', 'u2', 'u7', 'u100', 'Z1', 'u7'], ['tX', 'tY', 'tZ', 'tX', 'tW', 'tY'])
    # item overlap only (users prefixed like the ml-1m/ml-100k example)
    run('items',
        ['1m_1', '1m_2', '1m_10', '1m_3', '1m_1'], ['Toy Story (1995)', 'Heat (1995)', 'Jumanji (1995)', 'Heat (1995)', 'Zed (1999)'],
        ['100k_5', '100k_50', '100k_6', '100k_5'], ['Heat (1995)', 'Toy Story (1995)', 'Apollo 13 (1995)', 'Babe (1995)'])
    # empty overlap (OU == OI == 1)
    run('empty', ['a', 'b', 'c'], ['i1', 'i2', 'i3'], ['x', 'y'], ['j1', 'j2'])
    # both overlap + user_feat extending the user sets + unicode/utf-8 ordering + digits-vs-letters
    run('both_feat',
        ['é1', 'e2', 'z', '10', '9', 'A'], ['p', 'q', 'r', 'p', 'q', 's'],
        ['z', 'é1', '9', 'B', 'b'], ['q', 'r', 't', 'u', 'q'],
        s_user_feat=['é1', 'e2', 'z', '10', '9', 'A', 'onlyfeat_s', 'shared_feat'],
        t_user_feat=['z', 'é1', '9', 'B', 'b', 'shared_feat'])
    # NaN tokens on the source-only / target-only side are dropped (dataset.py:368-371,412-415)
    run('nan', ['a', np.nan, 'b', 'c'], ['i1', 'i2', 'i9', 'i3'], ['b', 'x', 'y'], ['i2', 'j1', np.nan])
    # larger random case
    rng = np.random.RandomState(0)
    toks_u = [f'u{n}' for n in rng.randint(0, 400, 600)]
    toks_i = [f'i{n}' for n in rng.randint(0, 300, 600)]
    toks_u2 = [f'u{n}' for n in rng.randint(200, 700, 500)]
    toks_i2 = [f'j{n}' for n in rng.randint(0, 300, 500)]
    run('random', toks_u, toks_i, toks_u2, toks_i2)


def gen_revoke_and_layout():
    out = {}
    # ---- revoke map (dataloader.py:240-247), called unbound
    class _S:
        pass
    fake = _S()
    user_num = 6
    fake.overlap_item_num = 5            # OI
    fake.revoke_item_num = 7             # TOI (num_target_only_item)
    fake.uid2positive_item = np.array([None] * user_num)
    fake.uid2history_item = np.array([None] * user_num)
    fake.uid2items_num = np.zeros(user_num, dtype=np.int64)
    cases = {1: ({1, 3, 12, 13, 20}, {3, 13}), 2: ({4, 12}, {4, 12}), 4: ({2, 19, 15, 14}, {19})}
    used_flat, used_ptr, pos_flat, pos_ptr = [], [0], [], [0]
    for uid in range(user_num):
        used, pos = cases.get(uid, (set(), set()))
        used_flat += sorted(used); used_ptr.append(len(used_flat))
        pos_flat += sorted(pos); pos_ptr.append(len(pos_flat))
        if uid in cases:
            CrossDomainFullSortEvalDataLoader._set_user_property(fake, uid, set(used), set(pos))
    out['revoke/OI'] = np.int64(5); out['revoke/TOI'] = np.int64(7)
    out['revoke/used_flat'] = np.array(used_flat, dtype=np.int64); out['revoke/used_ptr'] = np.array(used_ptr, dtype=np.int64)
    out['revoke/pos_flat'] = np.array(pos_flat, dtype=np.int64); out['revoke/pos_ptr'] = np.array(pos_ptr, dtype=np.int64)
    for uid in cases:
        out[f'revoke/positive/{uid}'] = np.sort(fake.uid2positive_item[uid].numpy())
        out[f'revoke/history/{uid}'] = np.sort(fake.uid2history_item[uid].numpy())
    out['revoke/items_num'] = fake.uid2items_num.copy()

    # ---- batch layout (dataloader.py:114-162): BOTH merges target<-source, source wraps, epoch len = target
    class _Loader:
        def __init__(self, name, n_batches, bs):
            self.name, self.n, self.bs, self.pr = name, n_batches, bs, 0
            self.pr_end = n_batches * bs
            self.served = []

        def __iter__(self):
            return self

        def __len__(self):
            return self.n

        def __next__(self):
            if self.pr >= self.pr_end:
                self.pr = 0
                raise StopIteration()
            b = self.pr // self.bs
            self.pr += self.bs
            self.served.append(b)
            return {f'{self.name}_user_id': torch.full((self.bs,), b, dtype=torch.int64)}

    trace = []
    fake = _S()
    fake.source_dataloader = _Loader('source', 2, 3)     # shorter than target: wraps
    fake.target_dataloader = _Loader('target', 5, 4)
    fake.overlap_dataloader = _Loader('overlap', 3, 2)
    fake._next_batch_data = lambda: CrossDomainDataloader._next_batch_data(fake)
    for state in (CrossDomainDataLoaderState.BOTH, CrossDomainDataLoaderState.SOURCE,
                  CrossDomainDataLoaderState.TARGET, CrossDomainDataLoaderState.OVERLAP):
        CrossDomainDataloader.set_mode(fake, state)
        n_len = CrossDomainDataloader.__len__(fake)
        ep = []
        while True:
            try:
                b = CrossDomainDataloader.__next__(fake)
            except StopIteration:
                break
            ep.append([int(b.get('source_user_id', torch.tensor([-1]))[0]),
                       int(b.get('target_user_id', torch.tensor([-1]))[0]),
                       int(b.get('overlap_user_id', torch.tensor([-1]))[0]),
                       len(b.get('source_user_id', [])), len(b.get('target_user_id', []))])
        out[f'layout/{state.name}/len'] = np.int64(n_len)
        out[f'layout/{state.name}/trace'] = np.array(ep, dtype=np.int64)
        # the reference leaves source.pr mid-way after a BOTH epoch only if it did not wrap exactly; it resets both
        out[f'layout/{state.name}/pr_after'] = np.array([fake.source_dataloader.pr, fake.target_dataloader.pr,
                                                         fake.overlap_dataloader.pr], dtype=np.int64)
    dump('revoke_layout', out)


if __name__ == '__main__':
    gen_emcdr()
    gen_cmf()
    gen_conet()
    gen_sscdr()
    gen_bitgcf()
    gen_clfm()
    gen_dtcdr()
    gen_deepapf()
    gen_natr()
    gen_dcdcsr()
    gen_remap()
    gen_revoke_and_layout()
