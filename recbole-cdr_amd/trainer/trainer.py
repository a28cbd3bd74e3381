"""CrossDomainTrainer: the caller of the hot path (recbole_cdr/trainer/trainer.py:18-76).

The reference subclasses third-party ``recbole.trainer.Trainer``; the part of it the phase loop relies on is restated
here (optimizer built ONCE and kept across phases, ``_train_epoch`` = zero_grad -> calculate_loss ->
(tuple => sum) -> nan check -> backward -> clip -> step, ``fit`` with eval_step / early stopping, ``evaluate`` =
full-sort scores -> mask PAD column and history -> top-k).  Metrics are computed on device with torch.topk (plumbing);
the dense optimizer is the native exact Adam (csrc/cdr_rows.hip ``cdr_adam_dense``).
"""
import numpy as np
import torch

from ..utils import train_mode2state, total_loss as _total


class DenseAdam(torch.optim.Optimizer):
    """torch.optim.Adam semantics (amsgrad=False), each parameter updated by one native kernel launch.  The step count is
    a device scalar bumped by a kernel, so ``step()`` is hipGraph-capturable (graph_step.GraphedTrainStep)."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        # (loss, loss_sum) device scalars handed over by the training loop before ``step()``: the launch that bumps the step counters
        # also adds this step's loss to the epoch's total (no launch of its own).  ``step`` sets it back to None once it has done so.
        self.loss_pair = None
        # cdr_batch_job's handed over by the training loop before ``step()`` (graph_step: the loader's NEXT batch): produced in workgroups
        # behind the update's own, in the same launch (cdr_adam_multi_dev_produce).  ``step`` sets it back to None once it has done so.
        self.produce_jobs = None

    @torch.no_grad()
    def step(self, closure=None):
        import ctypes
        from .. import binding as B_
        jobs, self.produce_jobs = self.produce_jobs, None
        try:
            self._step_groups(jobs)
        finally:
            jobs = self.__dict__.pop('_jobs_left', jobs)
            if jobs:                                   # no parameter had a gradient (or several groups): the batch is produced all the same
                from ..data.producer import launch_jobs
                launch_jobs(jobs)

    def _step_groups(self, jobs):
        import ctypes
        from .. import binding as B_
        self._jobs_left = jobs
        for group in self.param_groups:
            ps, gs, ms, vs, ns, ss, keep = [], [], [], [], [], [], []
            for p in group['params']:
                if p.grad is None:
                    continue                      # like torch: a parameter without a gradient keeps its own step count
                st = self.state[p]
                if not st:
                    st['step'] = torch.zeros(1, device=p.device, dtype=torch.int64)     # per-parameter, on device
                    st['exp_avg'] = torch.zeros_like(p)
                    st['exp_avg_sq'] = torch.zeros_like(p)
                g = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
                keep.append(g)
                ps.append(B_.f32(p.data)); gs.append(B_.f32(g)); ms.append(B_.f32(st['exp_avg'])); vs.append(B_.f32(st['exp_avg_sq']))
                ns.append(p.numel()); ss.append(B_.i64(st['step']))
                ps_dev = p.device
            if not ps:
                continue
            n = len(ps)
            arr = lambda xs: (ctypes.c_void_p * n)(*[x.value if hasattr(x, 'value') else x for x in xs])
            lp, self.loss_pair = self.loss_pair, None
            tk = self.__dict__.get('_ticket')
            if tk is None or tk.device != ps_dev:
                tk = self._ticket = torch.zeros(B_.SIGNIN_WORDS, device=ps_dev, dtype=torch.int32)        # sign-in words of the one-launch form
            jl = self._jobs_left
            if jl:
                self._jobs_left = None
                jarr = (B_.BatchJob * len(jl))(*jl)
                B_.call('cdr_adam_multi_dev_produce', B_.stream(), n, arr(ps), arr(gs), arr(ms), arr(vs), (ctypes.c_int64 * n)(*ns), arr(ss),
                        float(group['lr']), float(group['betas'][0]), float(group['betas'][1]), float(group['eps']),
                        float(group['weight_decay']), None if lp is None else B_.f32(lp[0]), None if lp is None else B_.f32(lp[1]), B_.raw(tk),
                        jarr, len(jl))
                continue
            B_.call('cdr_adam_multi_dev', B_.stream(), n, arr(ps), arr(gs), arr(ms), arr(vs), (ctypes.c_int64 * n)(*ns), arr(ss),
                    float(group['lr']), float(group['betas'][0]), float(group['betas'][1]), float(group['eps']),
                    float(group['weight_decay']), None if lp is None else B_.f32(lp[0]), None if lp is None else B_.f32(lp[1]), B_.raw(tk))


def _dense_adam_load(self, state_dict):
    """torch.optim.Adam checkpoints keep ``step`` as a Python float or a 0-d tensor: convert to this optimizer's device int64 [1]."""
    torch.optim.Optimizer.load_state_dict(self, state_dict)
    for p, st in self.state.items():
        if 'step' in st and not (torch.is_tensor(st['step']) and st['step'].dtype == torch.int64 and st['step'].numel() == 1
                                 and st['step'].dim() == 1):
            st['step'] = torch.full((1,), int(float(st['step'])), device=p.device, dtype=torch.int64)


DenseAdam.load_state_dict = _dense_adam_load


class RowAwareAdam(DenseAdam):
    """``DenseAdam`` for a model whose embedding tables take the deferred row-wise form of the SAME optimizer
    (``model.enable_deferred_adam``; lazyadam.DeferredRowAdam: exact dense-Adam results, O(batch) traffic).  ``step()`` = the
    dense kernels for every parameter that has a ``.grad`` (the MLP / mapping weights) + the row-wise update of the tables the
    model's backward left pending.  Drop-in for the reference's loop: zero_grad -> calculate_loss -> backward -> step."""

    def __init__(self, model, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        super().__init__(model.parameters(), lr=lr, betas=betas, eps=eps, weight_decay=weight_decay)
        self.row_opt = model.enable_deferred_adam(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay)

    @torch.no_grad()
    def step(self, closure=None):
        super().step(closure)
        self.row_opt.step()

    def on_replay(self):
        self.row_opt.on_replay()

    def state_dict(self):
        """The layout ``torch.optim.Adam.state_dict()`` writes -- the tables' flushed moments and update count under the standard
        per-parameter 'state' entries -- so that a checkpoint resumes under DenseAdam, torch.optim.Adam or this class alike
        (the moments used to sit under an extra key that the dense loaders silently ignored)."""
        sd = super().state_dict()
        rows = self.row_opt.state_dict()                                  # (flushes: every row at the current update)
        index, i = {}, 0
        for g in self.param_groups:
            for p_ in g['params']:
                index[id(p_)] = i
                i += 1
        for t, m, v in zip(self.row_opt.tables, rows['exp_avg'], rows['exp_avg_sq']):
            sd['state'][index[id(t)]] = {'step': torch.full((1,), int(rows['step']), device=t.device, dtype=torch.int64),
                                         'exp_avg': m, 'exp_avg_sq': v}
        return sd

    def load_state_dict(self, sd):
        sd = dict(sd)
        rows = sd.pop('deferred_rows', None)
        super().load_state_dict(sd)
        if rows is not None:                          # (checkpoints written before round 5 kept the tables' moments under this key)
            self.row_opt.load_state_dict(rows)
            return
        # a checkpoint written by DenseAdam / torch.optim.Adam: the tables' moments sit in the per-parameter state -- take them over
        # (a dense optimizer leaves every row at the same update count, which is what the row-wise form resumes from)
        ro = self.row_opt
        got = [self.state.get(t) for t in ro.tables]
        if all(st and 'exp_avg' in st for st in got):
            steps = {int(float(st['step'])) for st in got}
            if len(steps) != 1:
                raise ValueError('RowAwareAdam.load_state_dict: the tables carry different update counts (%s); the deferred row-wise '
                                 'form keeps ONE count for its tables' % sorted(steps))
            ro.load_state_dict({'step': steps.pop(), 'exp_avg': [st['exp_avg'] for st in got], 'exp_avg_sq': [st['exp_avg_sq'] for st in got]})
            for t in ro.tables:
                self.state.pop(t, None)
        elif any(st and 'exp_avg' in st for st in got):
            raise ValueError('RowAwareAdam.load_state_dict: optimizer state for only some of the embedding tables')


def early_stopping(value, best, cur_step, max_step, bigger=True):
    """recbole.utils.early_stopping."""
    stop_flag, update_flag = False, False
    if (bigger and value >= best) or (not bigger and value <= best):
        cur_step, best, update_flag = 0, value, True
    else:
        cur_step += 1
        if cur_step > max_step:
            stop_flag = True
    return best, cur_step, stop_flag, update_flag


class Trainer:
    def __init__(self, config, model):
        self.config, self.model = config, model
        self.learning_rate = config['learning_rate'] if 'learning_rate' in config else 1e-3
        self.epochs = int(config['epochs']) if 'epochs' in config else 1
        self.eval_step = min(config['eval_step'] if 'eval_step' in config else 1, self.epochs)
        self.stopping_step = config['stopping_step'] if 'stopping_step' in config else 10
        self.clip_grad_norm = config['clip_grad_norm'] if 'clip_grad_norm' in config else None
        self.valid_metric = (config['valid_metric'] if 'valid_metric' in config else 'MRR@10').lower()
        self.valid_metric_bigger = config['valid_metric_bigger'] if 'valid_metric_bigger' in config else True
        self.topk = config['topk'] if 'topk' in config else [10]
        self.device = config['device']
        self.weight_decay = config['weight_decay'] if 'weight_decay' in config else 0.0
        self.start_epoch, self.cur_step = 0, 0
        self.best_valid_score = -np.inf if self.valid_metric_bigger else np.inf
        self.best_valid_result = None
        self.train_loss_dict = dict()
        # 'dense'   : the reference's literal loop -- autograd into table-sized gradients + Adam over every parameter
        # 'rowwise' : the model's O(batch) fused step (tables too large for dense gradients, e.g. BASELINE config C5)
        self.optimizer_mode = config['optimizer_mode'] if 'optimizer_mode' in config else 'dense'
        on_gpu = torch.device(self.device).type == 'cuda'
        # The reference's Adam over every parameter.  A model that offers ``enable_deferred_adam`` (CoNet on its fused tower kernels)
        # gets the SAME optimizer with its embedding tables evaluated lazily per row -- bit-identical to the dense sweep
        # (lazyadam.py), O(batch) instead of O(table) traffic per step.  config['deferred_adam'] = False keeps the literal sweep;
        # gradient clipping needs the whole gradient, so it does too.
        deferred = (config['deferred_adam'] if 'deferred_adam' in config else True) and self.optimizer_mode == 'dense' and on_gpu \
            and not self.clip_grad_norm and hasattr(self.model, 'enable_deferred_adam') and getattr(self.model, 'fused_towers', True) \
            and all(p.is_cuda for p in self.model.parameters())
        if deferred:
            self.optimizer = RowAwareAdam(self.model, lr=self.learning_rate, weight_decay=self.weight_decay)
        else:
            self.optimizer = DenseAdam(self.model.parameters(), lr=self.learning_rate, weight_decay=self.weight_decay)
        # Dense mode on a GPU: each (phase, batch shape) is captured ONCE as a hipGraph -- batch production by the device loader,
        # calculate_loss, backward, optimizer -- and an epoch is a run of replays (graph_step.GraphedTrainStep); ragged tails and
        # models whose loss needs the host (``model.graph_key()`` is None) take the eager loop below.  config['graph_step'] = False
        # switches it off.
        self.graph_step = bool(config['graph_step'] if 'graph_step' in config else True) and on_gpu and self.optimizer_mode == 'dense' \
            and not self.clip_grad_norm
        # ... and rowwise mode on a device loader: {producer -> model.fused_train_step -> loss total} replayed where the model says its fused
        # step is capturable (model.fused_graph_key)
        self.graph_step_rowwise = bool(config['graph_step'] if 'graph_step' in config else True) and on_gpu and self.optimizer_mode == 'rowwise'
        # steps per graph launch on a device loader (the idle time between two graph launches is ~5-9 us: amortised over this many steps)
        # config['graph_unroll']; default 8 -- 16 for a step that runs software-pipelined over two streams (below): a two-stream graph's launch
        # boundary costs ~30-40 us of idle chip, 8 steps per graph leave ~4 us of it in every step (C3: 0.1535 -> 0.149 ms; 32: 0.1475;
        # the single-stream steps of C1 / C2 / C4 gain <= 1.5 % and keep 8 -- an epoch's tail of < unroll steps runs one step per launch)
        self.graph_unroll = int(config['graph_unroll']) if 'graph_unroll' in config else None
        # ... and, for models that can run part of the next step ahead (CoNet on the deferred Adam), those steps software-pipelined over two streams
        gp = config['graph_pipeline'] if 'graph_pipeline' in config else True
        self.graph_pipeline = gp if isinstance(gp, str) else bool(gp)          # True (one batch ahead) | 'two_ahead' | False
        self._graphs = {}
        self._loss_sum = None
        self.graph_stats = {'replayed': 0, 'eager': 0, 'captures': 0}
        # evaluation through the model's fused mask + top-k kernel when it has one (False: full score matrix + torch.topk)
        self.fused_topk = config['fused_topk'] if 'fused_topk' in config else True
        # config['dist_group'] (a torch.distributed group, or True for WORLD) with optimizer_mode='rowwise': the model's tables are
        # sharded over the group's GPUs; every rank runs the SAME loaders and trains on rows rank::world of every batch (a ragged
        # tail of < world rows is skipped, as a DistributedSampler(drop_last=True) would); evaluation is replicated.
        self.dist_group = None
        if 'dist_group' in config and config['dist_group'] not in (None, False):
            import torch.distributed as dist
            self.dist_group = dist.group.WORLD if config['dist_group'] is True else config['dist_group']
            if self.optimizer_mode != 'rowwise':
                raise ValueError("dist_group needs optimizer_mode='rowwise' (small-table models: dp.ShardedDataParallel)")
        if self.optimizer_mode not in ('dense', 'rowwise'):
            raise ValueError(f"optimizer_mode must be 'dense' or 'rowwise', got {self.optimizer_mode!r}")
        if self.optimizer_mode == 'rowwise' and not hasattr(self.model, 'fused_train_step'):
            raise NotImplementedError(f'{type(self.model).__name__} has no fused_train_step; use optimizer_mode=dense')
        if self.optimizer_mode == 'rowwise' and self.clip_grad_norm:
            # the fused step never materialises the gradient, so there is no norm to clip: refuse rather than ignore silently
            raise ValueError("clip_grad_norm is not supported with optimizer_mode='rowwise' (the fused step applies per-row updates "
                             "without forming the full gradient); use optimizer_mode='dense' or unset clip_grad_norm")
        if 'parallel_domains' in config and config['parallel_domains'] and self.dist_group is not None:
            import warnings
            warnings.warn('parallel_domains: the SOURCE and TARGET phases of a parallel stage run WITHOUT validation, early stopping, '
                          'best_valid_score updates and callback_fn (evaluation needs every rank); later phases evaluate as usual',
                          stacklevel=2)

    # ---- dense mode, one hipGraph replay per step --------------------------------------------------------------------------------
    def _graph_for(self, key, example=None, producer=None):
        """The captured step for ``key`` (built on first use; False once a capture has failed: that key stays eager)."""
        gs = self._graphs.get(key)
        if gs is None:
            from ..graph_step import GraphedTrainStep
            try:
                unroll = self.graph_unroll
                if unroll is None:
                    pipelined = (producer is not None and bool(self.graph_pipeline) and hasattr(self.model, 'prepare_batch')
                                 and hasattr(self.model, 'apply_rows_early') and getattr(self.optimizer, 'row_opt', None) is not None)
                    unroll = 16 if pipelined else 8
                gs = GraphedTrainStep(self.model, self.optimizer, example, producer=producer, loss_sum=self._loss_sum,
                                      unroll=unroll, pipeline=self.graph_pipeline)
                self.graph_stats['captures'] += 1
            except Exception as e:                                      # noqa: BLE001 -- reported, and the eager loop still trains
                import warnings
                back = 'parameters / optimizer state / loader counters restored to their values before the warm-up' \
                    if getattr(e, 'state_restored', False) else 'the warm-up steps that ran before the failure stay applied'
                warnings.warn(f'hipGraph capture of the training step failed for {key!r} ({type(e).__name__}: {e}); running it eagerly ({back})')
                gs = False
            self._graphs[key] = gs
        return gs

    def _eager_step(self, interaction):
        self.optimizer.zero_grad(set_to_none=True)
        losses = self.model.calculate_loss(interaction)
        loss = _total(losses)
        loss = loss.reshape(()) if loss.numel() == 1 else loss.sum()
        loss.backward()
        from ..graph_step import step_and_sum
        step_and_sum(self.optimizer, loss.detach(), self._loss_sum)
        self.graph_stats['eager'] += 1

    def _train_epoch_graphed(self, train_data, mkey):
        """recbole ``Trainer._train_epoch`` (zero_grad -> calculate_loss -> backward -> step per batch) with every full-shape batch
        served by ONE hipGraph replay.  With a device loader (``train_data.device_producer()``) the replay also produces the batch
        -- slice of the shuffled interactions, tiling, negative sampling -- so the host's part of a step is the replay call; the
        loader's own ``__next__`` serves what a capture cannot (ragged tails, the BOTH-mode source wrap) and ends the epoch."""
        if self._loss_sum is None:
            self._loss_sum = torch.zeros((), device=self.device, dtype=torch.float32)
        self._loss_sum.zero_()
        state = getattr(train_data, 'state', None)
        it = iter(train_data)                                  # (epoch shuffle; in place once a producer has pinned the columns)
        prod = train_data.device_producer() if hasattr(train_data, 'device_producer') else None
        if prod is not None:
            prod.resync()
            key = (mkey, state, 'device-loader')
            while True:
                gs = self._graphs.get(key)
                if prod.full_ahead() and gs is not False:
                    if gs is None:
                        gs = self._graph_for(key, producer=prod)
                        if gs is False:
                            prod.resync()
                            continue
                    k = gs.unroll
                    if k > 1 and prod.full_count() >= k:
                        gs.replay_many()                          # k steps, one graph launch
                        prod.advance(k)
                        self.graph_stats['replayed'] += k
                        continue
                    gs.replay()
                    prod.advance()
                    self.graph_stats['replayed'] += 1
                    continue
                try:
                    interaction = next(it)
                except StopIteration:
                    break
                self._eager_step(interaction)
                prod.resync()
        else:
            for interaction in it:
                interaction = interaction.to(self.device)
                sig = tuple((k, tuple(v.shape), str(v.dtype)) for k, v in interaction.items())
                key = (mkey, state, sig, getattr(interaction, 'k_major', None))
                gs = self._graphs.get(key)
                if gs is None and not any(k[:2] == key[:2] for k in self._graphs):
                    gs = self._graph_for(key, example=interaction)         # the first batch shape of this (model key, loader state)
                if gs and gs.matches(interaction):
                    gs.step(interaction)
                    self.graph_stats['replayed'] += 1
                else:
                    self._eager_step(interaction)                          # ragged tail (another shape), or a failed capture
        if self.graph_stats['replayed'] and hasattr(self.model, 'on_train_steps'):
            self.model.on_train_steps()                                    # replays ran no model Python: host-side caches are stale now
        value = float(self._loss_sum)
        if value != value:
            raise ValueError('Training loss is nan')
        return value

    def _train_epoch(self, train_data, epoch_idx):
        self.model.train()
        if self.graph_step and self.optimizer_mode == 'dense' and self.dist_group is None:
            mkey = self.model.graph_key() if hasattr(self.model, 'graph_key') else None
            if mkey is not None:
                return self._train_epoch_graphed(train_data, mkey)
        total = None                              # accumulated on device: no per-step host sync
        if self.optimizer_mode == 'rowwise' and self.dist_group is None and hasattr(train_data, 'device_producer'):
            prod = train_data.device_producer()
            if prod is not None:
                return self._train_epoch_rowwise_produced(train_data, prod)
        for interaction in train_data:
            interaction = interaction.to(self.device)
            if self.dist_group is not None:
                interaction = self._my_rows(interaction)
            if self.optimizer_mode == 'rowwise':
                loss = self.model.fused_train_step(interaction, lr=self.learning_rate, weight_decay=self.weight_decay)
                total = loss.detach().clone() if total is None else total + loss.detach()
                continue
            self.optimizer.zero_grad()
            losses = self.model.calculate_loss(interaction)
            loss = _total(losses)
            loss = loss.reshape(()) if loss.numel() == 1 else loss.sum()   # (a [1]-shaped loss: a view, not a reduction launch)
            loss.backward()
            if self.clip_grad_norm:
                torch.nn.utils.clip_grad_norm_(self.model.parameters(), **self.clip_grad_norm)
            self.optimizer.step()
            total = loss.detach() if total is None else total + loss.detach()
        value = float(total) if total is not None else 0.0
        if value != value:
            raise ValueError('Training loss is nan')
        return value

    def _train_epoch_rowwise_produced(self, train_data, prod):
        """The rowwise loop on a device loader: every full batch is ONE producer launch (slice of the shuffled interactions + tiling +
        negative sampling into fixed buffers, data/producer.py) followed by the model's fused O(batch) step; the loader's own
        ``__next__`` serves ragged tails and ends the epoch.  (The fused steps of large batches read host-side update counts, so
        they are launched, not replayed; at >= 1 M rows per step the host's part is < 2 % of the step.)"""
        if self._loss_sum is None:
            self._loss_sum = torch.zeros((), device=self.device, dtype=torch.float32)
        self._loss_sum.zero_()
        it = iter(train_data)
        prod.resync()
        # full batches of a capturable phase (model.fused_graph_key): the first two run eagerly on the capture stream -- real steps, they
        # create every buffer and context the capture needs -- then {producer -> fused step -> loss total} is replayed, 4 steps per launch
        gkey = self.model.fused_graph_key(prod.fields) if (self.graph_step_rowwise and hasattr(self.model, 'fused_graph_key')) else None
        gs = None
        if gkey is not None:
            gkey = (gkey, getattr(train_data, 'state', None), id(prod))
            gs = self._graphs.get(gkey)
            if gs is None:
                from ..graph_step import GraphedRowwiseStep
                gs = self._graphs[gkey] = GraphedRowwiseStep(self.model, prod, self._loss_sum,
                                                             dict(lr=self.learning_rate, weight_decay=self.weight_decay), unroll=4)
        while True:
            if gs is not None and gs is not False and prod.full_ahead():
                if gs.graph is None:
                    if gs.eager_steps < 2:
                        gs.eager(); prod.advance()
                        continue
                    try:
                        gs.capture()
                        self.graph_stats['captures'] += 1
                    except Exception as e:                              # noqa: BLE001 -- reported, and the eager loop still trains
                        import warnings
                        warnings.warn(f'hipGraph capture of the rowwise step failed for {gkey!r} ({type(e).__name__}: {e}); running it eagerly')
                        gs = self._graphs[gkey] = False
                        prod.resync()
                        continue
                many = gs.graph_k is not None and prod.full_count() >= gs.unroll
                gs.replay(many)
                prod.advance(gs.unroll if many else 1)
                self.graph_stats['replayed'] += gs.unroll if many else 1
                continue
            if prod.full_ahead():
                prod.launch()
                prod.advance()
                interaction = prod.fields
            else:
                try:
                    interaction = next(it)
                except StopIteration:
                    break
                prod.resync()
            loss = self.model.fused_train_step(interaction, lr=self.learning_rate, weight_decay=self.weight_decay)
            self._loss_sum.add_(loss.detach().reshape(()))
        value = float(self._loss_sum)
        if value != value:
            raise ValueError('Training loss is nan')
        return value

    def _my_rows(self, interaction):
        import torch.distributed as dist
        from ..data.interaction import Interaction
        if self.__dict__.get('_row_group') is not None:        # a domain's half of the ranks (parallel_domains): rows grank::half
            rank, world = self._row_group
        else:
            world, rank = dist.get_world_size(self.dist_group), dist.get_rank(self.dist_group)
        out = Interaction({k: v[:v.shape[0] - v.shape[0] % world][rank::world].contiguous() for k, v in interaction.items()})
        # rows j + m S of a k-major batch: every world-th row is again k-major (S / world positives) when world divides S
        # (EVERY field: in BOTH mode the source and target columns of one batch may differ in length)
        km = getattr(interaction, 'k_major', None)
        if km and all(v.shape[0] % km == 0 and (v.shape[0] // km) % world == 0 for v in interaction.values()):
            out.k_major = km
        return out

    def _topk_hits(self, interaction, n_user, history_index, positive_u, positive_i, kmax):
        """Hit matrix [U, kmax] and positives per user from the model's fused mask + top-k (no [U, N] score matrix): the
        history (rows, cols) pairs become a CSR with ascending columns, positives are matched by sorted (user, item) keys."""
        dev = self.device
        N = int(self.model.target_num_items) if hasattr(self.model, 'target_num_items') else None
        indptr = cols = None
        if history_index is not None:
            rows, hc = (torch.as_tensor(t, device=dev, dtype=torch.int64) for t in history_index)
            span = int(hc.max()) + 1 if hc.numel() else 1
            key = torch.sort(rows * span + hc).values
            cols = (key % span).contiguous()
            indptr = torch.zeros(n_user + 1, device=dev, dtype=torch.int64)
            indptr[1:] = torch.cumsum(torch.bincount(key // span, minlength=n_user), 0)
            if cols.numel() == 0:
                indptr = cols = None
        _vals, idx = self.model.full_sort_topk(interaction, kmax, hist_indptr=indptr, hist_cols=cols)
        pu = torch.as_tensor(positive_u, device=dev, dtype=torch.int64)
        pi = torch.as_tensor(positive_i, device=dev, dtype=torch.int64)
        span = max(int(idx.max()) + 1, int(pi.max()) + 1 if pi.numel() else 1, N or 1)
        pkey = torch.sort(pu * span + pi).values
        qkey = (torch.arange(n_user, device=dev).unsqueeze(1) * span + idx.clamp(min=0)).reshape(-1)
        at = torch.searchsorted(pkey, qkey).clamp(max=max(pkey.numel() - 1, 0))
        hit = (pkey[at] == qkey).view(n_user, kmax) & (idx >= 0) if pkey.numel() else torch.zeros_like(idx, dtype=torch.bool)
        return hit.float(), torch.bincount(pu, minlength=n_user).float()

    @torch.no_grad()
    def evaluate(self, eval_data):
        """eval_data yields (interaction, history_index (rows, cols) or None, positive_u, positive_i) like recbole's
        FullSortEvalDataLoader; returns {metric@k: value} for recall / mrr / ndcg / hit / precision."""
        self.model.eval()
        kmax = max(self.topk)
        fused = self.fused_topk and hasattr(self.model, 'full_sort_topk')
        # recbole cuts the evaluated users so that the [U, N] score matrix fits eval_batch_size entries (ONE user per call over a 10 M-item
        # catalogue at the default 4,096).  The fused mask + top-k path never forms that matrix, so it re-cuts a loader that allows it
        # into throughput-sized batches (config['eval_users_per_batch'], default 1,024; 0 keeps recbole's cut): same users, same metrics
        want = int(self.config['eval_users_per_batch']) if 'eval_users_per_batch' in self.config else 1024
        restore = None
        if fused and want > 0 and hasattr(eval_data, 'rebatch') and getattr(eval_data, 'step', want) < want:
            restore = eval_data.step                      # (the caller's cut comes back afterwards: another consumer may need [U, N] to fit)
            eval_data.rebatch(want)
        # the model's evaluation-mode caches (CoNet: the item half of layer 1, the packed one-user call) live exactly as long as this loop:
        # nothing trains inside it
        frozen = hasattr(self.model, 'freeze_for_eval')
        if frozen:
            self.model.freeze_for_eval()
        try:
            return self._evaluate_batches(eval_data, fused, kmax)
        finally:
            if frozen:
                self.model.unfreeze_eval()
            if restore is not None:
                eval_data.rebatch(restore)

    def _evaluate_batches(self, eval_data, fused, kmax):
        hits, pos_len = [], []
        for interaction, history_index, positive_u, positive_i in eval_data:
            interaction = interaction.to(self.device)
            n_user = len(interaction)
            if fused:
                h, npos = self._topk_hits(interaction, n_user, history_index, positive_u, positive_i, kmax)
                hits.append(h); pos_len.append(npos)
                continue
            scores = self.model.full_sort_predict(interaction).view(n_user, -1)
            scores[:, 0] = -np.inf
            if history_index is not None:
                scores[history_index] = -np.inf
            _, topk_idx = torch.topk(scores, kmax, dim=-1)
            pos = torch.zeros_like(scores, dtype=torch.bool)
            pos[positive_u, positive_i] = True
            hits.append(torch.gather(pos, 1, topk_idx).float())
            pos_len.append(pos.sum(1).float())
        hit = torch.cat(hits)
        n_pos = torch.cat(pos_len).clamp(min=1)
        result = {}
        ranks = torch.arange(1, kmax + 1, device=hit.device, dtype=torch.float32)
        for k in self.topk:
            h = hit[:, :k]
            result[f'recall@{k}'] = float((h.sum(1) / n_pos).mean())
            result[f'hit@{k}'] = float((h.sum(1) > 0).float().mean())
            result[f'precision@{k}'] = float((h.sum(1) / k).mean())
            first = (h * (1.0 / ranks[:k])).max(1).values
            result[f'mrr@{k}'] = float(first.mean())
            dcg = (h / torch.log2(ranks[:k] + 1)).sum(1)
            ideal = torch.cumsum(1.0 / torch.log2(ranks[:k] + 1), 0)
            idcg = ideal[(n_pos.clamp(max=k).long() - 1)]
            result[f'ndcg@{k}'] = float((dcg / idcg).mean())
        return result

    def save_checkpoint(self, path, epoch=None):
        """recbole ``Trainer._save_checkpoint``: config-free subset -- epoch, early-stopping state, model ``state_dict``,
        ``other_parameter`` and the optimizer state (dense: ``DenseAdam.state_dict()``; rowwise: the model's per-table
        moments and update counts).  With a ``dist_group``: one file per rank, ``<path>.rank<r>``."""
        if self.dist_group is not None:
            # a sharded model: every rank writes ITS shards (tables + moments in their current layout) next to the replicated parts
            import torch.distributed as dist
            state = {'epoch': epoch, 'cur_step': self.cur_step, 'best_valid_score': self.best_valid_score,
                     'phase': getattr(self.model, 'phase', None), 'dist': self.model.dist_checkpoint()}
            torch.save(state, f'{path}.rank{dist.get_rank(self.dist_group)}')
            return
        state = {'epoch': epoch, 'cur_step': self.cur_step, 'best_valid_score': self.best_valid_score,
                 'state_dict': self.model.state_dict(), 'other_parameter': self.model.other_parameter(),
                 'optimizer_mode': self.optimizer_mode, 'optimizer': self.optimizer.state_dict(),
                 'phase': getattr(self.model, 'phase', None)}
        if self.optimizer_mode == 'rowwise' and hasattr(self.model, 'fused_optimizer_state'):
            state['rowwise'] = self.model.fused_optimizer_state()
        torch.save(state, path)

    def resume_checkpoint(self, path):
        """recbole ``Trainer.resume_checkpoint``."""
        if self.dist_group is not None:
            import torch.distributed as dist
            state = torch.load(f'{path}.rank{dist.get_rank(self.dist_group)}', map_location=self.device, weights_only=False)
            self.start_epoch = (state['epoch'] + 1) if state['epoch'] is not None else 0
            self.cur_step, self.best_valid_score = state['cur_step'], state['best_valid_score']
            self.model.load_dist_checkpoint(state['dist'])
            if state.get('phase') is not None:
                self.model.set_phase(state['phase'])
            return
        state = torch.load(path, map_location=self.device, weights_only=False)
        self.start_epoch = (state['epoch'] + 1) if state['epoch'] is not None else 0
        self.cur_step, self.best_valid_score = state['cur_step'], state['best_valid_score']
        self._graphs.clear()                       # captured steps point at the optimizer state tensors about to be replaced
        self.model.load_state_dict(state['state_dict'])
        self.model.load_other_parameter(state.get('other_parameter'))
        if state.get('phase') is not None and hasattr(self.model, 'set_phase'):
            self.model.set_phase(state['phase'])
        self.optimizer.load_state_dict(state['optimizer'])
        if 'rowwise' in state and hasattr(self.model, 'load_fused_optimizer_state'):
            self.model.load_fused_optimizer_state(state['rowwise'])

    def fit(self, train_data, valid_data=None, verbose=True, saved=True, show_progress=False, callback_fn=None):
        for epoch_idx in range(self.start_epoch, self.epochs):
            train_loss = self._train_epoch(train_data, epoch_idx)
            self.train_loss_dict[epoch_idx] = train_loss
            if self.eval_step <= 0 or not valid_data:
                continue
            if (epoch_idx + 1) % self.eval_step == 0:
                valid_result = self.evaluate(valid_data)
                valid_score = valid_result[self.valid_metric]
                self.best_valid_score, self.cur_step, stop_flag, update_flag = early_stopping(
                    valid_score, self.best_valid_score, self.cur_step, max_step=self.stopping_step,
                    bigger=self.valid_metric_bigger)
                if update_flag:
                    self.best_valid_result = valid_result
                if callback_fn:
                    callback_fn(epoch_idx, valid_score)
                if stop_flag:
                    break
        return self.best_valid_score, self.best_valid_result


class CrossDomainTrainer(Trainer):
    """Phase loop over ``train_modes`` (SOURCE / TARGET / BOTH / OVERLAP) -- trainer.py:43-76."""

    def __init__(self, config, model):
        super().__init__(config, model)
        self.train_modes = config['train_modes']
        self.train_epochs = config['epoch_num']
        self.split_valid_flag = config['source_split']

    def _reinit(self, phase):
        """Reset per-phase state; the optimizer (and its Adam moments) is NOT rebuilt (trainer.py:30-41, Q13)."""
        self.start_epoch = 0
        self.cur_step = 0
        self.best_valid_score = -np.inf if self.valid_metric_bigger else np.inf
        self.best_valid_result = None
        self.item_tensor = None
        self.tot_item_num = None
        self.train_loss_dict = dict()
        self.epochs = int(self.train_epochs[phase])
        self.eval_step = min(self.config['eval_step'] if 'eval_step' in self.config else 1, self.epochs)

    def _fit_domains_in_parallel(self, train_data, phase, both, verbose, saved, show_progress, callback_fn):
        """config['parallel_domains'] with a dist_group: a SOURCE phase followed by a TARGET phase (or the reverse) touches
        disjoint tables and disjoint optimizer state, so the lower half of the ranks runs the SOURCE epochs while the upper half
        runs the TARGET epochs -- each domain's tables cut into world/2 column slices instead of world (twice the slice width,
        half the replicated index work: DESIGN.md 6.1).  Same result as running the two phases one after the other; no
        evaluation inside (it needs every rank); the halves meet again at a barrier.  ``both`` False: a SOURCE or TARGET phase
        on its own -- only that domain's half trains, the other half waits."""
        import torch.distributed as dist
        world, rank = dist.get_world_size(self.dist_group), dist.get_rank(self.dist_group)
        half = world // 2
        mine = 'SOURCE' if rank < half else 'TARGET'
        schemes = [self.train_modes[phase]] + ([self.train_modes[phase + 1]] if both else [])
        if hasattr(self.model, 'dist_prepare'):
            self.model.dist_prepare([s_.lower() for s_ in schemes])          # collective: every rank, also the idle half
        if mine in schemes:
            self._reinit(phase + schemes.index(mine))
            train_data.set_mode(train_mode2state[mine])
            self.model.set_phase(mine)
            self._row_group = (rank % half, half)
            try:
                Trainer.fit(self, train_data, None, verbose, saved, show_progress, callback_fn)
            finally:
                self._row_group = None
        dist.barrier(group=self.dist_group)

    def _fit_domains_on_two_streams(self, train_data, phase):
        """config['parallel_domains'] on ONE GPU (optimizer_mode='rowwise'): a SOURCE phase directly followed by a TARGET phase (or the
        reverse) touches disjoint tables and disjoint optimizer state, so their epochs are enqueued side by side on two HIP streams --
        the tails and small launches of one domain's step run under the other domain's kernels (measured on the headline step: 4.27 ->
        3.99 ms for a source + target step).  Bit-identical to running the two phases one after the other; no evaluation inside (the
        caller falls back to the sequential phases when validation is requested).  Returns {scheme: [epoch loss sums]}."""
        from ..data.producer import DeviceBatchProducer
        schemes = [self.train_modes[phase], self.train_modes[phase + 1]]
        epochs = {sc: int(self.train_epochs[phase + j]) for j, sc in enumerate(schemes)}
        loaders = {'SOURCE': train_data.source_dataloader, 'TARGET': train_data.target_dataloader}
        cache = self.__dict__.setdefault('_two_stream', {})
        if 'streams' not in cache:
            cache['streams'] = {sc: torch.cuda.Stream(device=self.device) for sc in ('SOURCE', 'TARGET')}
            cache['loss'] = {sc: torch.zeros((), device=self.device, dtype=torch.float32) for sc in ('SOURCE', 'TARGET')}
            cache['prod'] = {}
        streams, sums = cache['streams'], cache['loss']
        for sc in schemes:
            if sc not in cache['prod']:
                cache['prod'][sc] = DeviceBatchProducer(loaders[sc]) if DeviceBatchProducer.supports(loaders[sc]) else None
        prods = cache['prod']
        cur = torch.cuda.current_stream()
        log = {sc: [] for sc in schemes}
        self.model.train()
        for e in range(max(epochs.values())):
            active = [sc for sc in schemes if e < epochs[sc]]
            its = {}
            for sc in active:
                streams[sc].wait_stream(cur)
                with torch.cuda.stream(streams[sc]):
                    its[sc] = iter(loaders[sc])                          # (the epoch shuffle, on the domain's own stream)
                    if prods[sc] is not None:
                        prods[sc].resync()
                    sums[sc].zero_()
            live = list(active)
            while live:
                for sc in list(live):
                    with torch.cuda.stream(streams[sc]):
                        pr = prods[sc]
                        if pr is not None and pr.full_ahead():
                            pr.launch(); pr.advance()
                            batch = pr.fields
                        else:
                            try:
                                batch = next(its[sc]).to(self.device)
                            except StopIteration:
                                live.remove(sc)
                                continue
                            if pr is not None:
                                pr.resync()
                        self.model.set_phase(sc)                          # (host-side switch: which tables the fused step takes)
                        loss = self.model.fused_train_step(batch, lr=self.learning_rate, weight_decay=self.weight_decay)
                        sums[sc].add_(loss.detach().reshape(()))
            for sc in active:
                cur.wait_stream(streams[sc])
            for sc in active:
                v = float(sums[sc])
                if v != v:
                    raise ValueError('Training loss is nan')
                log[sc].append(v)
        for ld in loaders.values():
            ld.pr = 0
        self.train_loss_dict = {sc: dict(enumerate(v)) for sc, v in log.items()}
        self.model.set_phase(schemes[-1])
        return log

    def fit(self, train_data, valid_data=None, verbose=True, saved=True, show_progress=False, callback_fn=None):
        parallel = False
        if self.dist_group is not None and 'parallel_domains' in self.config and self.config['parallel_domains']:
            import torch.distributed as dist
            parallel = dist.get_world_size(self.dist_group) % 2 == 0
        two_streams = (self.dist_group is None and 'parallel_domains' in self.config and bool(self.config['parallel_domains'])
                       and self.optimizer_mode == 'rowwise' and torch.device(self.device).type == 'cuda'
                       and (valid_data is None or self.config['eval_step'] <= 0 if 'eval_step' in self.config else valid_data is None))
        skip = False
        for phase in range(len(self.train_modes)):
            if skip:                                 # ran together with the previous phase
                skip = False
                continue
            nxt = self.train_modes[phase + 1] if phase + 1 < len(self.train_modes) else None
            if two_streams and {self.train_modes[phase], nxt} == {'SOURCE', 'TARGET'}:
                self._reinit(phase)
                self._fit_domains_on_two_streams(train_data, phase)
                skip = True
                continue
            if parallel and self.train_modes[phase] in ('SOURCE', 'TARGET'):
                both = {self.train_modes[phase], nxt} == {'SOURCE', 'TARGET'}
                self._fit_domains_in_parallel(train_data, phase, both, verbose, saved, show_progress, callback_fn)
                skip = both
                continue
            self._reinit(phase)
            scheme = self.train_modes[phase]
            train_data.set_mode(train_mode2state[scheme])
            self.model.set_phase(scheme)
            if self.split_valid_flag and valid_data is not None:
                source_valid_data, target_valid_data = valid_data
                if scheme == 'SOURCE':
                    super().fit(train_data, source_valid_data, verbose, saved, show_progress, callback_fn)
                else:
                    super().fit(train_data, target_valid_data, verbose, saved, show_progress, callback_fn)
            else:
                super().fit(train_data, valid_data, verbose, saved, show_progress, callback_fn)
        self.model.set_phase('OVERLAP')
        return self.best_valid_score, self.best_valid_result


class DCDCSRTrainer(CrossDomainTrainer):
    """trainer.py:79-139: the same phase loop, except that the BOTH phase (DCDCSR's mapping phase: the loss ignores the batch and
    nothing can be ranked before the second TARGET visit builds the affine table) runs WITHOUT validation data.
    ``train_modes`` typically ['SOURCE', 'TARGET', 'BOTH', 'TARGET'] (properties/model/DCDCSR.yaml): ``epoch_num`` is indexed by the
    phase's POSITION, and the model counts its visits of each phase (dcdcsr.py:90-92)."""

    def fit(self, train_data, valid_data=None, verbose=True, saved=True, show_progress=False, callback_fn=None):
        for phase in range(len(self.train_modes)):
            self._reinit(phase)
            scheme = self.train_modes[phase]
            train_data.set_mode(train_mode2state[scheme])
            self.model.set_phase(scheme)
            if scheme == 'BOTH':
                Trainer.fit(self, train_data, None, verbose, saved, show_progress, callback_fn)
            elif self.split_valid_flag and valid_data is not None:
                source_valid_data, target_valid_data = valid_data
                Trainer.fit(self, train_data, source_valid_data if scheme == 'SOURCE' else target_valid_data, verbose, saved,
                            show_progress, callback_fn)
            else:
                Trainer.fit(self, train_data, valid_data, verbose, saved, show_progress, callback_fn)
        self.model.set_phase('OVERLAP')
        return self.best_valid_score, self.best_valid_result
