"""The whole path at scale, through the reference's own loop: synthetic cross-domain dataset (resident on the device) ->
four-state loader with the DEVICE negative sampler -> CrossDomainTrainer(optimizer_mode='rowwise') over SOURCE, TARGET and
OVERLAP epochs (EMCDR-BPR, D=128: FusedBPRStep / FusedMapStep on the model's own tables) -> full-sort evaluation with the
fused mask + top-k kernel.  Reports wall-clock interactions/s per phase INCLUDING sampling, batching and Python.

Over several GPUs: ``python -m torch.distributed.run --nproc-per-node N --master-addr 127.0.0.1 tools/e2e_train.py`` (one rank per
GPU; config['dist_group'] = True: tables sharded, every rank trains rows rank::N of each batch, evaluation replicated).
CDR_BENCH_SHARED_GPU=1 puts every rank on cuda:0 over gloo -- a functional check on a one-GPU box, not a measurement."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import recbole_cdr_amd  # noqa: F401
from recbole_cdr_amd.data import CrossDomainDataloader, OverlapDataloader, DomainTrainLoader, FullSortEvalLoader
from recbole_cdr_amd.data.synthetic import SyntheticCrossDomainDataset
from recbole_cdr_amd.model.cross_domain_recommender.emcdr import EMCDR
from recbole_cdr_amd.sampler import DeviceNegSampler
from recbole_cdr_amd.trainer import CrossDomainTrainer
from recbole_cdr_amd.utils import InputType

WORLD, RANK = int(os.environ.get('WORLD_SIZE', '1')), int(os.environ.get('RANK', '0'))
SHARED = bool(int(os.environ.get('CDR_BENCH_SHARED_GPU', '0')))
dev = 'cuda:%d' % (0 if SHARED else int(os.environ.get('LOCAL_RANK', '0')))
torch.cuda.set_device(dev)
if WORLD > 1:
    import torch.distributed as dist
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29544')
    dist.init_process_group('gloo' if SHARED else 'nccl', rank=RANK, world_size=WORLD)
    if RANK:
        sys.stdout = open(os.devnull, 'w')                # rank 0 reports; every rank computes the same numbers
OU, TOI, NI, BATCH = int(os.environ.get('E2E_USERS', 4_000_001)), int(os.environ.get('E2E_ITEMS', 1_000_000)), \
    int(os.environ.get('E2E_INTER', 12_000_000)), 1 << 20
t0 = time.time()
ds = SyntheticCrossDomainDataset(OU=OU, TOU=0, SOU=0, OI=1, TOI=TOI, SOI=TOI, n_source_inter=NI, n_target_inter=NI)
C = int(os.environ.get('E2E_CLUSTERS', 0))
EPOCHS = os.environ.get('E2E_EPOCHS', '2')
if C:
    # learnable structure: user u and item i interact only when u % C == (item's index in its domain) % C
    rng = np.random.RandomState(7)
    def clustered(n, item_lo):
        u = rng.randint(1, OU, n)
        j = rng.randint(0, TOI // C, n) * C + u % C                       # index inside the domain, same residue as the user
        return np.unique(np.stack([u, item_lo + np.minimum(j, TOI - 1)], 1), axis=0)
    ds.t_pairs, ds.s_pairs = clustered(NI, 1), clustered(NI, 1 + TOI)
    rng.shuffle(ds.t_pairs)
cfg = {'source_domain': {'NEG_PREFIX': 'neg_'}, 'target_domain': {'NEG_PREFIX': 'neg_'}, 'device': dev,
       'latent_factor_model': 'BPR', 'source_embedding_size': 128, 'target_embedding_size': 128, 'reg_weight': 0.01,
       'mapping_function': 'linear', 'mlp_hidden_size': [128], 'learning_rate': float(os.environ.get('E2E_LR', 1e-3)), 'optimizer_mode': 'rowwise',
       'train_modes': ['SOURCE', 'TARGET', 'OVERLAP'], 'epoch_num': [EPOCHS, EPOCHS, '2'], 'source_split': False, 'eval_step': 0,
       'epochs': int(EPOCHS), 'learning_rate_note': 'lr below', 'topk': [10], 'valid_metric': 'Recall@10'}
if WORLD > 1:
    cfg['dist_group'] = True
    cfg['parallel_domains'] = bool(int(os.environ.get('E2E_PARALLEL', '1')))      # SOURCE and TARGET epochs on their own halves of the ranks
torch.manual_seed(2022)                                   # the same seed on every rank: replicated loaders and samplers draw alike
model = EMCDR(cfg, ds).to(dev)
dt = lambda a: torch.from_numpy(a.copy()).to(dev)
held = 20_000                                           # target interactions held out for the evaluation
t_tr, t_te = ds.t_pairs[held:], ds.t_pairs[:held]
s_smp, t_smp = DeviceNegSampler(ds, 'source', ds.s_pairs, dev), DeviceNegSampler(ds, 'target', ds.t_pairs, dev)
train = CrossDomainDataloader(
    DomainTrainLoader({'source_user_id': dt(ds.s_pairs[:, 0]), 'source_item_id': dt(ds.s_pairs[:, 1])}, 'source_user_id', 'source_item_id',
                      'source_label', 'neg_', BATCH, 1, InputType.PAIRWISE, s_smp, shuffle=True),
    DomainTrainLoader({'target_user_id': dt(t_tr[:, 0]), 'target_item_id': dt(t_tr[:, 1])}, 'target_user_id', 'target_item_id',
                      'target_label', 'neg_', BATCH, 1, InputType.PAIRWISE, t_smp, shuffle=True),
    OverlapDataloader(OU, 65536, device=dev, shuffle=True))
print(f'setup (host-side synthetic data, CSR for the sampler, tables): {time.time() - t0:.1f} s; '
      f'{len(ds.s_pairs)} source / {len(t_tr)} target interactions, {OU - 1} overlapped users, {TOI} items per domain', flush=True)
trainer = CrossDomainTrainer(cfg, model)
orig, log = trainer._train_epoch, []
def timed(data, e):
    torch.cuda.synchronize(); t = time.time()
    v = orig(data, e)
    torch.cuda.synchronize(); log.append((time.time() - t, v))
    return v
trainer._train_epoch = timed
trainer.fit(train)
rows = [('SOURCE', len(ds.s_pairs))] * int(EPOCHS) + [('TARGET', len(t_tr))] * int(EPOCHS) + [('OVERLAP', OU)] * 2   # first epoch of a phase
if WORLD > 1 and cfg['parallel_domains'] and WORLD % 2 == 0:
    rows = [r for r in rows if r[0] != 'TARGET']                # rank 0 reports: it trained the SOURCE domain while the upper half trained TARGET
for (phase, n), (sec, loss) in zip(rows, log):                                                  # also builds its step objects
    print(f'{phase:8s} epoch: {sec * 1e3:9.1f} ms wall for {n} rows = {n / sec / 1e6:8.1f} M rows/s (epoch loss sum {loss:.4f})', flush=True)
# evaluation in the target domain after the OVERLAP phase (users mapped through the learned mapping): fused mask + top-10
te_users = np.unique(t_te[:, 0])[:4096]
t_te = t_te[np.isin(t_te[:, 0], te_users)]
hist = t_tr[np.isin(t_tr[:, 0], te_users)]
loader = FullSortEvalLoader('target_user_id', t_te, hist, ds.num_overlap_item + ds.num_target_only_item, 1024 * (1 + TOI), dev)
if C:
    model.set_phase('TARGET')
    r = trainer.evaluate(loader)
    print(f'clustered data ({C} clusters): recall@10 in the TARGET phase {r["recall@10"]:.5f}, hit@10 {r["hit@10"]:.5f} '
          f'(random ranking: {10.0 / TOI:.6f}; a perfect cluster model: ~{min(1.0, 10.0 * C / TOI):.4f})', flush=True)
    model.set_phase('OVERLAP')
for attempt in ('first call (allocates the workspaces)', 'second call'):
    torch.cuda.synchronize(); t = time.time()
    res = trainer.evaluate(loader)
    torch.cuda.synchronize(); sec = time.time() - t
    print(f'evaluate, {attempt}: {len(te_users)} users x {1 + TOI} items in {sec * 1e3:.1f} ms = '
          f'{len(te_users) * (1 + TOI) / sec / 1e9:.1f} G items/s wall; recall@10 {res["recall@10"]:.5f}', flush=True)
