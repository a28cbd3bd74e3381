"""EMCDR on libcdrhip -- same class contract as recbole_cdr/model/cross_domain_recommender/emcdr.py.

  calculate_source_loss / calculate_target_loss (emcdr.py:110-154): ONE fused gather-dot-loss launch per step
      (MF: cdr_point_fwd MSE ; BPR: cdr_bpr_fwd) instead of ~8 ATen kernels, EmbLoss rows shared with the score rows.
  calculate_map_loss (emcdr.py:156-168): 2 row gathers + fp32-MFMA linear layers + MSE.
  predict / full_sort_predict (emcdr.py:178-233): mapped-or-target select (K7) + fp32-MFMA scoring over <= 2 row
      ranges of the item table (no torch.cat copy).
Parameter names equal the reference's, so its checkpoints' ``state_dict`` load unchanged.
"""
import torch
import torch.nn as nn

from ... import binding as B_
from ... import functional as F_
from ...utils import InputType
from ..crossdomain_recommender import CrossDomainRecommender, xavier_normal_initialization


class EMCDR(CrossDomainRecommender):

    def __init__(self, config, dataset):
        super().__init__(config, dataset)
        self.mode = self.one_sided_overlap_mode()
        self.phase = 'both'

        self.latent_factor_model = config['latent_factor_model']
        if self.latent_factor_model == 'MF':
            self.input_type = InputType.POINTWISE
            self.SOURCE_LABEL = dataset.source_domain_dataset.label_field
            self.TARGET_LABEL = dataset.target_domain_dataset.label_field
        else:
            self.input_type = InputType.PAIRWISE
        self.source_latent_dim = config['source_embedding_size']
        self.target_latent_dim = config['target_embedding_size']
        self.reg_weight = config['reg_weight']
        self.map_func = config['mapping_function']
        if self.map_func == 'linear':
            self.mapping = nn.Linear(self.source_latent_dim, self.target_latent_dim, bias=False)
        else:
            assert config["mlp_hidden_size"] is not None
            dims = [self.source_latent_dim] + list(config["mlp_hidden_size"]) + [self.target_latent_dim]
            self.mapping = self.mlp_layers(dims)

        # union-sized tables (emcdr.py:67-71).  The reference's zero-fill of "foreign" rows is overwritten by its own
        # xavier init, so no rows are zeroed here either.
        self.source_user_embedding = nn.Embedding(self.total_num_users, self.source_latent_dim)
        self.source_item_embedding = nn.Embedding(self.total_num_items, self.source_latent_dim)
        self.target_user_embedding = nn.Embedding(self.total_num_users, self.target_latent_dim)
        self.target_item_embedding = nn.Embedding(self.total_num_items, self.target_latent_dim)
        self.bpr_gamma = 1e-10
        self.apply(xavier_normal_initialization)
        # optional: the tables sharded over the GPUs of a node (optimizer_mode='rowwise' only; see _dist_train_step)
        self.__dict__['_dist_cfg'] = config['dist_group'] if 'dist_group' in config else None
        self.__dict__['_dist_parallel'] = bool(config['parallel_domains']) if 'parallel_domains' in config else False

    @staticmethod
    def mlp_layers(layer_dims):
        mods = []
        for i, (d_in, d_out) in enumerate(zip(layer_dims[:-1], layer_dims[1:])):
            mods.append(nn.Linear(d_in, d_out))
            if i != len(layer_dims[:-1]) - 1:
                mods.append(nn.Tanh())
        return nn.Sequential(*mods)

    def set_phase(self, phase):
        self.phase = phase

    def graph_key(self):
        return None if self._dist_group() is not None else ('EMCDR', self.phase)

    # ---- mapping function on the fp32 MFMA kernel --------------------------------------------------------------
    def mapping_layers(self):
        """[(weight, bias or None, activation after the layer)] of the mapping function (emcdr.py:59-64,86-93)."""
        if self.map_func == 'linear':
            return [(self.mapping.weight, None, B_.ACT_NONE)]
        lins = [m for m in self.mapping if isinstance(m, nn.Linear)]
        return [(l.weight, l.bias, B_.ACT_TANH if n != len(lins) - 1 else B_.ACT_NONE) for n, l in enumerate(lins)]

    def apply_mapping(self, x):
        if self.map_func == 'linear':
            return F_.linear(x, self.mapping.weight, None, B_.ACT_NONE)
        layers = [m for m in self.mapping if isinstance(m, nn.Linear)]
        for n, lin in enumerate(layers):
            act = B_.ACT_TANH if n != len(layers) - 1 else B_.ACT_NONE
            x = F_.linear(x, lin.weight, lin.bias, act)
        return x

    # ---- losses -------------------------------------------------------------------------------------------------
    def _domain_loss(self, interaction, domain):
        U = getattr(self, f'{domain}_user_embedding').weight
        I = getattr(self, f'{domain}_item_embedding').weight
        user = interaction[getattr(self, f'{domain.upper()}_USER_ID')]
        item = interaction[getattr(self, f'{domain.upper()}_ITEM_ID')]
        if self.latent_factor_model == 'MF':
            label = interaction[getattr(self, f'{domain.upper()}_LABEL')]
            loss, _ = F_.PointGatherLoss.apply(B_.CDR_LOSS_MSE, U, I, None, None, user, item, label, self.reg_weight)
            return loss
        neg = interaction[getattr(self, f'{domain.upper()}_NEG_ITEM_ID')]
        return F_.BPRGatherLoss.apply(U, I, user, item, neg, self.bpr_gamma, self.reg_weight)

    def calculate_source_loss(self, interaction):
        return self._domain_loss(interaction, 'source')

    def calculate_target_loss(self, interaction):
        return self._domain_loss(interaction, 'target')

    def calculate_map_loss(self, interaction):
        idx = interaction[self.OVERLAP_ID]                      # [OB,1]
        kind = 'user' if self.mode == 'overlap_users' else 'item'
        src = F_.gather_rows(getattr(self, f'source_{kind}_embedding').weight, idx)
        tgt = F_.gather_rows(getattr(self, f'target_{kind}_embedding').weight, idx)
        return F_.mse_loss(self.apply_mapping(src), tgt)

    def calculate_loss(self, interaction):
        self._whole_tables_only('calculate_loss')
        if self.phase == 'SOURCE':
            return self.calculate_source_loss(interaction)
        elif self.phase == 'OVERLAP':
            return self.calculate_map_loss(interaction)
        else:
            return self.calculate_target_loss(interaction)

    # ---- O(batch) training step (large tables) ------------------------------------------------------------------
    def fused_train_step(self, interaction, opt='adam', lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        """``calculate_loss -> backward -> optimizer.step`` of the current phase without table-sized gradients or a dense
        optimizer sweep (fused.FusedBPRStep / fused.FusedMapStep on this model's own tables): what
        ``CrossDomainTrainer`` runs when ``config['optimizer_mode'] == 'rowwise'``.  Same loss and per-row gradients as
        ``calculate_loss``; the embedding tables take the row-wise (lazy) Adam, the mapping function the exact dense one.
        One optimizer state per table, shared by the phases.  Both latent factor models (MF: pointwise MSE; BPR)."""
        from ...fused import FusedBPRStep, FusedPointStep, FusedMapStep, RowwiseState, OPT_ADAM, OPT_SGD
        code = OPT_ADAM if opt == 'adam' else OPT_SGD
        cache = self.__dict__.setdefault('_fused', {'states': {}, 'steps': {}})
        if self._dist_group() is not None:
            return self._dist_train_step(interaction, code, dict(opt=opt, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))

        def state(name):
            if name not in cache['states']:
                cache['states'][name] = RowwiseState(getattr(self, name).weight.data, code)
            return cache['states'][name]

        hp = dict(opt=opt, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay)
        if self.phase == 'OVERLAP':
            kind = 'user' if self.mode == 'overlap_users' else 'item'
            idx = interaction[self.OVERLAP_ID]
            key = ('map', kind)
            if key not in cache['steps']:
                cache['steps'][key] = FusedMapStep(
                    getattr(self, f'source_{kind}_embedding').weight.data, getattr(self, f'target_{kind}_embedding').weight.data,
                    self.apply_mapping, list(self.mapping.parameters()), idx.numel(), layers=self.mapping_layers(),
                    source_state=state(f'source_{kind}_embedding'), target_state=state(f'target_{kind}_embedding'), **hp)
                pending = self.__dict__.get('_pending_map_state', {}).pop(kind, None)
                if pending is not None and cache['steps'][key].map_opt is not None:
                    cache['steps'][key].map_opt.load_state_dict(pending)
            # the reference's OverlapDataloader yields slices of a shuffled arange (data/dataloader.py:37-52): distinct ids, which
            # the two-launch step relies on.  A caller feeding its own, possibly repeated ids sets model.overlap_ids_unique = False.
            return cache['steps'][key].step(idx, unique=getattr(self, 'overlap_ids_unique', True))
        domain = 'source' if self.phase == 'SOURCE' else 'target'
        user = interaction[getattr(self, f'{domain.upper()}_USER_ID')].reshape(-1)
        item = interaction[getattr(self, f'{domain.upper()}_ITEM_ID')].reshape(-1)
        if self.latent_factor_model == 'MF':
            label = interaction[getattr(self, f'{domain.upper()}_LABEL')].reshape(-1).float()
            # recbole's pointwise batches tile S positives 1 + k times (Interaction.point_k, set by the loader): at large batches the
            # per-positive step gathers each user row once and updates it in place (in this layout NO user row occurs once per row list)
            pk = getattr(interaction, 'point_k', None)
            rows = user.numel()
            D_ = getattr(self, f'{domain}_user_embedding').weight.shape[1]
            if pk is not None and 1 <= pk <= 64 and rows % (1 + pk) == 0 and rows > 8192 and D_ % 4 == 0 and D_ <= 256:
                from ...fused import KMajorPointStep
                key = ('mfk', domain, pk)
                step = cache['steps'].get(key)
                if step is None or step.max_positives < rows // (1 + pk):
                    step = KMajorPointStep(getattr(self, f'{domain}_user_embedding').weight.data,
                                           getattr(self, f'{domain}_item_embedding').weight.data, rows // (1 + pk), k=pk, loss='mse',
                                           reg_weight=self.reg_weight, user_state=state(f'{domain}_user_embedding'),
                                           item_state=state(f'{domain}_item_embedding'), **hp)
                    cache['steps'][key] = step
                return step.step(user, item, label)[0]
            key = ('mf', domain)
            step = cache['steps'].get(key)
            if step is None or step.max_batch < user.numel():
                step = FusedPointStep(getattr(self, f'{domain}_user_embedding').weight.data,
                                      getattr(self, f'{domain}_item_embedding').weight.data, user.numel(), loss='mse',
                                      reg_weight=self.reg_weight, user_state=state(f'{domain}_user_embedding'),
                                      item_state=state(f'{domain}_item_embedding'), **hp)
                cache['steps'][key] = step
            return step.step(user, item, label)[0]
        neg = interaction[getattr(self, f'{domain.upper()}_NEG_ITEM_ID')].reshape(-1)
        # recbole's pairwise batches tile S positives k times with k-major negatives (crossdomain_sampler.py:148-152); the loader
        # says so (Interaction.k_major).  Then the per-positive step serves k >= 2 (u, p gathered once per positive) and every
        # batch small enough for its four-launch form (the reference default of 2,048 rows); k = 1 at large batches stays on the
        # per-triple step, which is faster there.
        k = getattr(interaction, 'k_major', None)
        rows = user.numel()
        if k is not None and rows % k == 0 and (k >= 2 or rows + rows // k <= 8192):
            from ...fused import KMajorBPRStep
            key = ('bprk', domain, k)
            step = cache['steps'].get(key)
            if step is None or step.max_positives < rows // k:
                step = KMajorBPRStep(getattr(self, f'{domain}_user_embedding').weight.data,
                                     getattr(self, f'{domain}_item_embedding').weight.data, rows // k, k=k, gamma=self.bpr_gamma,
                                     reg_weight=self.reg_weight, user_state=state(f'{domain}_user_embedding'),
                                     item_state=state(f'{domain}_item_embedding'), **hp)
                cache['steps'][key] = step
            return step.step(user, item, neg)[0]
        key = ('bpr', domain)
        step = cache['steps'].get(key)
        if step is None or step.max_batch < user.numel():
            step = FusedBPRStep(getattr(self, f'{domain}_user_embedding').weight.data,
                                getattr(self, f'{domain}_item_embedding').weight.data, user.numel(), gamma=self.bpr_gamma,
                                reg_weight=self.reg_weight, user_state=state(f'{domain}_user_embedding'),
                                item_state=state(f'{domain}_item_embedding'), **hp)
            cache['steps'][key] = step
        return step.step(user, item, neg)[0]

    def fused_graph_key(self, interaction):
        """Hashable tag of the launches ``fused_train_step(interaction)`` would make, or None when they must not be captured in a
        hipGraph: capturable are the per-triple BPR step (update counts on the device: cdr_bpr_step_fused_dev) and the distinct-id
        OVERLAP step; the MF steps and the per-positive forms read host-side update counts."""
        if self._dist_group() is not None:
            return None
        if self.phase == 'OVERLAP':
            return ('map', self.mode) if getattr(self, 'overlap_ids_unique', True) and self.map_func in ('linear', 'non_linear') else None
        if self.latent_factor_model == 'MF':
            return None
        domain = 'source' if self.phase == 'SOURCE' else 'target'
        k = getattr(interaction, 'k_major', None)
        rows = interaction[getattr(self, f'{domain.upper()}_USER_ID')].numel()
        if k is not None and rows % k == 0 and (k >= 2 or rows + rows // k <= 8192):
            return None                                        # per-positive forms (fused.KMajorBPRStep)
        if rows > 65536:
            # Above ~200 k keys rocPRIM sorts with its Onesweep configuration, whose temporary-storage resets do not survive a hipGraph
            # replay on this ROCm (the same defect cdr_common.h records for hipMemsetAsync inside a captured step): the replayed sort
            # hands the applies garbage positions -- an illegal access at B = 1,048,576, found by bench.py --only-e2e.  Nothing is lost:
            # at these sizes the step is the sum of its kernels, a replay saves no time.
            return None
        return ('bpr', domain, rows)

    def fused_replayed(self, n=1):
        """Host bookkeeping of ``n`` hipGraph replays of the current phase's ``fused_train_step`` (the update counts' host mirrors)."""
        cache = self.__dict__.get('_fused', {'steps': {}})
        if self.phase == 'OVERLAP':
            kind = 'user' if self.mode == 'overlap_users' else 'item'
            st = cache['steps'].get(('map', kind))
            for _ in range(n):
                st.sstate.advance(device_bumped=True)
                st.tstate.advance(device_bumped=True)
            return
        domain = 'source' if self.phase == 'SOURCE' else 'target'
        cache['steps'][('bpr', domain)].replayed(n)

    # ---- the same step over the GPUs of a node (config['dist_group']: a torch.distributed group, or True for WORLD) ----------
    _TABLES = ('source_user_embedding', 'source_item_embedding', 'target_user_embedding', 'target_item_embedding')

    def _whole_tables_only(self, what):
        if self.__dict__.get('_dist') is not None:
            raise RuntimeError(f'EMCDR.{what} needs whole tables, but this model holds one shard of each (config["dist_group"]): '
                               'use fused_train_step / full_sort_topk, or gather_full_tables() into a single-process model')

    def _dist_group(self):
        g = self.__dict__.get('_dist_cfg')
        if g is None or g is False:
            return None
        import torch.distributed as dist
        return dist.group.WORLD if g is True else g

    def _dist_tables(self, code):
        """First use: make every rank's parameters rank 0's, then keep only this rank's shard of each embedding table (the
        nn.Embedding weights become views of the shards: column slices [rows, D/G] while the BPR / MF phases train, row shards in
        the OVERLAP phase -- dimshard.ShardedTables)."""
        T = self.__dict__.get('_dist')
        if T is None:
            import torch.distributed as dist
            from ...dimshard import ShardedTables
            grp = self._dist_group()
            T = ShardedTables(grp, code)
            src = dist.get_global_rank(grp, 0)
            for p in self.parameters():
                dist.broadcast(p.data, src, group=grp)
            # parallel_domains: the SOURCE and the TARGET phase touch disjoint tables, so each domain's tables live (in the
            # dimension layout) on one half of the ranks and the trainer runs the two phases at the same time
            halves = None
            if self.__dict__.get('_dist_parallel') and T.world >= 2 and T.world % 2 == 0:
                half = T.world // 2
                to_global = lambda rs: [dist.get_global_rank(grp, r) for r in rs]
                halves = {'source': (list(range(half)), dist.new_group(to_global(range(half)))),
                          'target': (list(range(half, T.world)), dist.new_group(to_global(range(half, T.world))))}
            for name in self._TABLES:
                emb = getattr(self, name)
                holders, hgroup = halves[name.split('_')[0]] if halves else (None, None)
                shard = T.adopt(name, emb.weight.data, 'dim', holders=holders, hgroup=hgroup)
                emb.weight.data = shard if shard is not None else emb.weight.data.new_empty(0, emb.weight.shape[1])
            self.__dict__['_dist'] = T
        return T

    def _dist_state(self, T, name, layout):
        st = T.state(name, layout)
        if st is None:
            raise RuntimeError(f'this rank holds no part of {name} in the {layout!r} layout: with parallel_domains the SOURCE and '
                               'TARGET phases run on their own halves of the ranks (CrossDomainTrainer schedules them so)')
        getattr(self, name).weight.data = st.table             # the module's parameter IS the shard the kernels update
        return st

    def dist_prepare(self, domains):
        """Collective over the whole group: bring the tables of ``domains`` ('source' / 'target') into the dimension layout before
        a phase that only part of the ranks will train (the transposes involve every rank, also those that end up holding
        nothing of a table)."""
        from ...fused import OPT_ADAM
        T = self._dist_tables(OPT_ADAM)
        for name in self._TABLES:
            if name.split('_')[0] in domains:
                st = T.state(name, 'dim')
                emb = getattr(self, name)
                emb.weight.data = st.table if st is not None else emb.weight.data.new_empty(0, emb.weight.shape[1])

    def _dist_train_step(self, interaction, code, hp):
        """SOURCE / TARGET: dimshard.DimShardedBPRStep / DimShardedPointStep on this rank's column slices, fed with this rank's
        rows of the batch (the same count on every rank).  OVERLAP: both tables (and their moments) are transposed to row shards
        once, then fused.FusedMapStep(group=...) exchanges nothing but ids.  Returns the GLOBAL batch's loss on every rank."""
        from ...dimshard import DimShardedBPRStep, DimShardedPointStep
        from ...fused import FusedMapStep
        T = self._dist_tables(code)
        cache = self.__dict__['_fused']
        if self.phase == 'OVERLAP':
            kind = 'user' if self.mode == 'overlap_users' else 'item'
            names = (f'source_{kind}_embedding', f'target_{kind}_embedding')
            sst, tst = (self._dist_state(T, n, 'row') for n in names)
            key = ('map', kind, T.version(names[0]), T.version(names[1]))
            step = cache['steps'].get(key)
            if step is None:
                for k in [k for k in cache['steps'] if k[:2] == key[:2]]:
                    del cache['steps'][k]                         # a step object of an earlier layout: its buffers can go
                step = FusedMapStep(sst.table, tst.table, self.apply_mapping, list(self.mapping.parameters()), 1, group=T.group,
                                    source_state=sst, target_state=tst, **hp)
                if cache.get('map_opt') is not None:
                    step.map_opt = cache['map_opt']               # the mapping's Adam state outlives a layout change
                pending = self.__dict__.pop('_pending_dist_map_state', None)
                if pending is not None and step.map_opt is not None:
                    step.map_opt.load_state_dict(pending)         # ... and a checkpoint
                cache['map_opt'] = step.map_opt
                cache['steps'][key] = step
            for n in names:
                T.touched(n)
            return step.step(interaction[self.OVERLAP_ID])
        domain = 'source' if self.phase == 'SOURCE' else 'target'
        names = (f'{domain}_user_embedding', f'{domain}_item_embedding')
        ust, ist = (self._dist_state(T, n, 'dim') for n in names)
        user = interaction[getattr(self, f'{domain.upper()}_USER_ID')].reshape(-1)
        item = interaction[getattr(self, f'{domain.upper()}_ITEM_ID')].reshape(-1)
        if user.numel() == 0:                      # the trainer's ragged-tail rule left nothing (on every rank alike)
            return torch.zeros((), device=user.device, dtype=torch.float32)
        mf = self.latent_factor_model == 'MF'
        key = ('mf' if mf else 'bpr', domain, T.version(names[0]), T.version(names[1]))
        step = cache['steps'].get(key)
        if step is None or step.max_batch < user.numel() * step.world:
            for k in [k for k in cache['steps'] if k[:2] == key[:2]]:
                del cache['steps'][k]
            if mf:
                step = DimShardedPointStep(ust.table, ist.table, user.numel(), loss='mse', reg_weight=self.reg_weight,
                                           group=T.holder_group(names[0]), user_state=ust, item_state=ist, **hp)
            else:
                step = DimShardedBPRStep(ust.table, ist.table, user.numel(), gamma=self.bpr_gamma, reg_weight=self.reg_weight,
                                         group=T.holder_group(names[0]), user_state=ust, item_state=ist, **hp)
            cache['steps'][key] = step
        for n in names:
            T.touched(n)
        third = (interaction[getattr(self, f'{domain.upper()}_LABEL')].reshape(-1).float() if mf
                 else interaction[getattr(self, f'{domain.upper()}_NEG_ITEM_ID')].reshape(-1))
        return step.step(user, item, third)[0]

    @torch.no_grad()
    def _dist_full_sort_topk(self, interaction, k, hist_indptr, hist_cols):
        """``full_sort_topk`` over row shards: every rank evaluates the same users against ITS item rows (fused mask + top-k
        kernel), k candidates per user are all-gathered and merged -- identical result on every rank (shard.ShardedFullSort)."""
        from ...shard import ShardedFullSort
        from ...fused import OPT_ADAM
        T = self._dist_tables(OPT_ADAM)
        OI, TI = self.overlapped_num_items, self.target_num_items
        if self.phase == 'SOURCE':                  # emcdr.py:208-214: cat(W_s[:OI], W_s[TI:]) -- two row ranges of the sharded table
            fs = ShardedFullSort(T.rows('source_item_embedding'), self.total_num_items, group=T.group)
            user_e = fs.user_rows(T.rows('source_user_embedding'), interaction[self.SOURCE_USER_ID])
            return fs.topk_ranges(user_e, k, [(0, OI), (TI, self.total_num_items)], hist_indptr=hist_indptr, hist_cols=hist_cols,
                                  exclude_first_col=True)
        user = interaction[self.TARGET_USER_ID]
        items = T.rows('target_item_embedding')
        if self.phase != 'TARGET' and self.mode != 'overlap_users':
            # item-overlap: rows [0, OI) of the scored slab are mapping(source rows); both tables follow the same r % G rule, so a
            # rank maps exactly the local rows it owns
            n_l = len(range(T.rank, OI, T.world))
            items = items.clone()
            if n_l:
                items[:n_l] = self.apply_mapping(T.rows('source_item_embedding')[:n_l].contiguous())
        fs = ShardedFullSort(items, TI, group=T.group)
        user_e = fs.user_rows(T.rows('target_user_embedding'), user)
        if self.phase != 'TARGET' and self.mode == 'overlap_users':
            mapped = self.apply_mapping(fs.user_rows(T.rows('source_user_embedding'), user))
            user_e = torch.where((user < self.overlapped_num_users).unsqueeze(1), mapped, user_e)      # emcdr.py:219-226 (Q5)
        return fs.topk(user_e, k, hist_indptr=hist_indptr, hist_cols=hist_cols, exclude_first_col=True)

    def dist_checkpoint(self):
        """This rank's part of the sharded model for a checkpoint: per table its layout, holder ranks, update count and the
        local table / moments (None where the rank holds nothing); the replicated mapping and its dense Adam state."""
        T = self.__dict__.get('_dist')
        if T is None:
            raise RuntimeError('no sharded state yet: nothing was trained or evaluated in distributed mode')
        tabs = {}
        for name, e in T.entries.items():
            st = e['state']
            tabs[name] = {'layout': e['layout'], 'holders': list(e['holders']), 'step': st.step if st is not None else None,
                          'table': st.table if st is not None else None,
                          'exp_avg': st.exp_avg if st is not None else None, 'exp_avg_sq': st.exp_avg_sq if st is not None else None}
        mo = self.__dict__.get('_fused', {}).get('map_opt')
        return {'world': T.world, 'rank': T.rank, 'tables': tabs, 'mapping': self.mapping.state_dict(),
                'map_opt': mo.state_dict() if mo is not None else None}

    def load_dist_checkpoint(self, state):
        """Restore what ``dist_checkpoint`` returned on the same rank of a group of the same size (collective: a table saved in the
        row layout is first brought there on every rank)."""
        from ...fused import OPT_ADAM
        T = self._dist_tables(OPT_ADAM)
        if (state['world'], state['rank']) != (T.world, T.rank):
            raise ValueError(f"checkpoint of rank {state['rank']}/{state['world']} loaded on rank {T.rank}/{T.world}")
        for name, rec in state['tables'].items():
            e = T.entries[name]
            if list(rec['holders']) != list(e['holders']):
                raise ValueError(f'{name}: saved with holder ranks {rec["holders"]}, this run uses {e["holders"]} (parallel_domains differs)')
            st = T.state(name, rec['layout'])
            if st is not None:
                st.step = int(rec['step'])
                st.table.copy_(rec['table'])
                if st.exp_avg is not None and rec['exp_avg'] is not None:
                    st.exp_avg.copy_(rec['exp_avg']); st.exp_avg_sq.copy_(rec['exp_avg_sq'])
            emb = getattr(self, name)
            emb.weight.data = st.table if st is not None else emb.weight.data.new_empty(0, emb.weight.shape[1])
            e['version'] += 1                                     # step objects built on the old tensors are stale
            e['eval_rows'] = None
        self.mapping.load_state_dict(state['mapping'])
        cache = self.__dict__.setdefault('_fused', {'states': {}, 'steps': {}})
        cache['steps'].clear()
        cache['map_opt'] = None
        self.__dict__['_pending_dist_map_state'] = state.get('map_opt')

    def gather_full_tables(self):
        """{name: replicated [rows, D] table} from the shards (checkpoint / hand-over to a single-process model)."""
        T = self.__dict__.get('_dist')
        return {n: T.full(n) for n in self._TABLES} if T is not None else {n: getattr(self, n).weight.data for n in self._TABLES}

    def fused_optimizer_state(self):
        """Row-wise optimizer state of ``fused_train_step`` for a checkpoint: per table the moments and the update count, plus
        the dense Adam state of the mapping function (recbole's checkpoint stores ``optimizer.state_dict()``; this is its
        counterpart for ``optimizer_mode='rowwise'``)."""
        cache = self.__dict__.get('_fused')
        if not cache:
            return {}
        out = {'tables': {k: {'step': st.step, 'exp_avg': st.exp_avg, 'exp_avg_sq': st.exp_avg_sq} for k, st in cache['states'].items()}}
        for key, step in cache['steps'].items():
            if key[0] == 'map' and step.map_opt is not None:
                out.setdefault('mapping', {})[key[1]] = step.map_opt.state_dict()
        return out

    def load_fused_optimizer_state(self, state, opt='adam'):
        """Restore what ``fused_optimizer_state`` returned (before the next ``fused_train_step``)."""
        from ...fused import RowwiseState, OPT_ADAM, OPT_SGD
        cache = self.__dict__.setdefault('_fused', {'states': {}, 'steps': {}})
        code = OPT_ADAM if opt == 'adam' else OPT_SGD
        for name, rec in state.get('tables', {}).items():
            st = cache['states'].get(name)
            if st is None:
                st = cache['states'][name] = RowwiseState(getattr(self, name).weight.data, code)
            st.step = int(rec['step'])
            if rec['exp_avg'] is not None:
                st.exp_avg.copy_(rec['exp_avg']); st.exp_avg_sq.copy_(rec['exp_avg_sq'])
        self._pending_map_state = state.get('mapping', {})

    # ---- scoring ------------------------------------------------------------------------------------------------
    def _mapped_rows(self, kind, ids, n_overlap):
        """where(id < n_overlap, mapping(source[id]), target[id]); the mapping is evaluated for all rows (Q5)."""
        src = F_.gather_rows(getattr(self, f'source_{kind}_embedding').weight, ids)
        return F_.select_mapped(self.apply_mapping(src), getattr(self, f'target_{kind}_embedding').weight, ids, n_overlap)

    @staticmethod
    def _rowdot(a, b):
        # [B,D].[B,D] -> [B] as the diagonal-free pointwise kernel: reuse the scoring GEMM on row pairs would be
        # O(B^2); B is the eval batch of explicit (user,item) pairs, so gather-dot it natively.
        n, D = a.shape
        ids = torch.arange(n, device=a.device, dtype=torch.int64)
        zeros = torch.zeros(n, device=a.device, dtype=torch.float32)
        _, scores = F_.PointGatherLoss.apply(B_.CDR_LOSS_MSE, a.contiguous(), b.contiguous(), None, None, ids, ids, zeros, 0.0)
        return scores

    def _pair_scores(self, U, I, user, item):
        zeros = torch.zeros(user.numel(), device=U.device, dtype=torch.float32)
        _, scores = F_.PointGatherLoss.apply(B_.CDR_LOSS_MSE, U, I, None, None, user, item, zeros, 0.0)
        return scores

    @torch.no_grad()
    def predict(self, interaction):
        self._whole_tables_only('predict')
        if self.phase == 'SOURCE':
            return self._pair_scores(self.source_user_embedding.weight, self.source_item_embedding.weight,
                                     interaction[self.SOURCE_USER_ID], interaction[self.SOURCE_ITEM_ID])
        user, item = interaction[self.TARGET_USER_ID], interaction[self.TARGET_ITEM_ID]
        if self.phase == 'TARGET':
            return self._pair_scores(self.target_user_embedding.weight, self.target_item_embedding.weight, user, item)
        if self.mode == 'overlap_users':
            user_e = self._mapped_rows('user', user, self.overlapped_num_users)
            item_e = F_.gather_rows(self.target_item_embedding.weight, item)
        else:
            user_e = F_.gather_rows(self.target_user_embedding.weight, user)
            item_e = self._mapped_rows('item', item, self.overlapped_num_items)
        return self._rowdot(user_e, item_e)

    @torch.no_grad()
    def _full_sort_operands(self, interaction):
        """(user_e [U,D], slab0, slab1): the operands of the all-items contraction of the current phase; the item operand is
        <= 2 contiguous row ranges, so the reference's torch.cat copies (emcdr.py:212-214,228-230) never happen."""
        OI, TI = self.overlapped_num_items, self.target_num_items
        if self.phase == 'SOURCE':
            user_e = F_.gather_rows(self.source_user_embedding.weight, interaction[self.SOURCE_USER_ID])
            W = self.source_item_embedding.weight
            return user_e, W[:OI], W[TI:]
        if self.phase == 'TARGET':
            user_e = F_.gather_rows(self.target_user_embedding.weight, interaction[self.TARGET_USER_ID])
            return user_e, self.target_item_embedding.weight[:TI], None
        user = interaction[self.TARGET_USER_ID]
        if self.mode == 'overlap_users':
            return self._mapped_rows('user', user, self.overlapped_num_users), self.target_item_embedding.weight[:TI], None
        user_e = F_.gather_rows(self.target_user_embedding.weight, user)
        return user_e, self.apply_mapping(self.source_item_embedding.weight[:OI]), self.target_item_embedding.weight[OI:TI]

    @torch.no_grad()
    def full_sort_predict(self, interaction):
        self._whole_tables_only('full_sort_predict')
        user_e, slab0, slab1 = self._full_sort_operands(interaction)
        return F_.fullsort_scores(user_e, slab0 if slab0.shape[0] else None, slab1).view(-1)

    @torch.no_grad()
    def full_sort_topk(self, interaction, k, hist_indptr=None, hist_cols=None):
        """Evaluation without the [U, N] matrix: (values [U,k], columns [U,k]) of ``full_sort_predict`` after recbole's
        mask (column 0 and the per-user history columns, CSR with ascending columns) -- what ``Trainer.evaluate`` needs."""
        if self._dist_group() is not None:
            return self._dist_full_sort_topk(interaction, k, hist_indptr, hist_cols)
        user_e, slab0, slab1 = self._full_sort_operands(interaction)
        return F_.fullsort_topk(user_e, slab0 if slab0.shape[0] else None, slab1, k=k, hist_indptr=hist_indptr,
                                hist_cols=hist_cols, exclude_first_col=True)
