"""CoNet on libcdrhip -- same class contract as recbole_cdr/model/cross_domain_recommender/conet.py:25-242.

Every (user, item) row runs BOTH towers (the cross term of one tower needs the other tower's activations,
conet.py:118-137); per layer that is four fp32-MFMA contractions, the two cross products accumulated in place before
the ReLU with the overlap mask as a per-row scale (cdr_gemm_f32_ex) -- no boolean-mask indexing, hence none of the
reference's 104 `nonzero` host syncs per step.  Quirks kept: PAD id 0 counts as
overlapped; the regulariser is the UN-weighted sum of ||H_l||_F (reg_weight is unused); predict returns [B,1];
full_sort_predict returns [U,N] from the target tower without cross terms.
"""
import torch
import torch.nn as nn

from ... import binding as B_
from ... import functional as F_
from ...utils import InputType
from ..crossdomain_recommender import CrossDomainRecommender, xavier_normal_initialization


class CoNet(CrossDomainRecommender):
    input_type = InputType.POINTWISE

    def __init__(self, config, dataset):
        super().__init__(config, dataset)
        self.SOURCE_LABEL = dataset.source_domain_dataset.label_field
        self.TARGET_LABEL = dataset.target_domain_dataset.label_field
        self.mode = self.one_sided_overlap_mode()
        self.latent_dim = config['embedding_size']
        self.reg_weight = config['reg_weight']
        self.cross_layers = list(config["mlp_hidden_size"])

        self.source_user_embedding = nn.Embedding(self.total_num_users, self.latent_dim)
        self.target_user_embedding = nn.Embedding(self.total_num_users, self.latent_dim)
        self.source_item_embedding = nn.Embedding(self.total_num_items, self.latent_dim)
        self.target_item_embedding = nn.Embedding(self.total_num_items, self.latent_dim)

        dims = [2 * self.latent_dim] + self.cross_layers
        self.source_crossunit_linear = nn.ModuleList([nn.Linear(a, b) for a, b in zip(dims[:-1], dims[1:])])
        self.source_outputunit = nn.Sequential(nn.Linear(self.cross_layers[-1], 1), nn.Sigmoid())
        self.target_crossunit_linear = nn.ModuleList([nn.Linear(a, b) for a, b in zip(dims[:-1], dims[1:])])
        self.target_outputunit = nn.Sequential(nn.Linear(self.cross_layers[-1], 1), nn.Sigmoid())
        self.crossparas = nn.ModuleList([nn.Linear(a, b, bias=False) for a, b in zip(dims[:-1], dims[1:])])
        self.apply(xavier_normal_initialization)
        # layer widths the fused tower kernels take (multiples of 4, <= 8 layers, fits LDS); otherwise -- or with
        # config['conet_fused'] = False -- every cross unit is four launches of the generic MFMA GEMM (_towers)
        self._dims = tuple(dims)
        fused = config['conet_fused'] if 'conet_fused' in config else True
        self.fused_towers = bool(fused) and self.latent_dim % 4 == 0 and F_.conet_supported(self._dims)
        self.row_opt = None          # lazyadam.DeferredRowAdam over the four tables (enable_deferred_adam)

    # ---- exact dense Adam over the four tables, evaluated lazily per row (lazyadam.py) ---------------------------------------
    def table_parameters(self):
        return [self.source_user_embedding.weight, self.source_item_embedding.weight, self.target_user_embedding.weight,
                self.target_item_embedding.weight]

    def enable_deferred_adam(self, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        """The reference's torch.optim.Adam semantics for the embedding tables at O(batch) cost per step: rows without a
        gradient postpone their momentum updates and replay them when next read -- bit-identical to the dense sweep
        (tests: test_deferred_adam_*).  Needs the fused tower path.  Returns the optimizer (``.step()`` after backward;
        trainer.RowAwareAdam does that inside ``optimizer.step()``)."""
        from ...lazyadam import DeferredRowAdam
        assert self.fused_towers, 'the deferred row-wise Adam rides on the fused tower kernels'
        self.row_opt = DeferredRowAdam(self.table_parameters(), [0, 1, 0, 1], lr=lr, betas=betas, eps=eps, weight_decay=weight_decay)
        return self.row_opt

    def train(self, mode=True):
        if mode:
            self.__dict__.pop('_eval_frozen', None)
            self._drop_eval_cache()
        return super().train(mode)

    def _drop_eval_cache(self):
        """full_sort_predict's evaluation-mode caches: the item part of layer 1 and the packed few-users call built on it."""
        self.__dict__.pop('_eval_P', None)
        self.__dict__.pop('_eval_few', None)

    # The caches are only ever built and used between ``freeze_for_eval()`` and ``unfreeze_eval()`` -- the caller's promise that no
    # parameter changes in between (CrossDomainTrainer.evaluate brackets its loop of full_sort_predict calls with the pair).  Outside such
    # a bracket every call recomputes from the live parameters, as the reference does (conet.py:171-181): an in-place parameter change under
    # model.eval() -- an optimizer step, p.data.copy_(), an EMA swap, a native update through raw pointers -- is seen by the next call.
    def freeze_for_eval(self):
        self._drop_eval_cache()
        self.__dict__['_eval_frozen'] = True
        return self

    def unfreeze_eval(self):
        self.__dict__.pop('_eval_frozen', None)
        self._drop_eval_cache()
        return self

    def _eval_caching(self):
        return (not self.training) and bool(self.__dict__.get('_eval_frozen', False))

    def __getstate__(self):
        # copy.deepcopy / pickle / torch.save(model): the caches hold ctypes pointers (not picklable) and are cheap to rebuild
        st = dict(self.__dict__)
        for k in ('_eval_P', '_eval_few', '_eval_frozen'):
            st.pop(k, None)
        return st

    def on_train_steps(self):
        self.__dict__.pop('_eval_frozen', None)
        self._drop_eval_cache()

    def load_state_dict(self, *args, **kwargs):
        self.__dict__.pop('_eval_frozen', None)
        self._drop_eval_cache()
        return super().load_state_dict(*args, **kwargs)

    def sync_tables(self):
        """Every table row up to date (no-op unless a deferred optimizer has postponed work): before anything reads whole tables."""
        if self.row_opt is not None:
            self.row_opt.flush()

    # ---- both towers through every cross unit ----------------------------------------------------------------------
    def _towers(self, user, item):
        self.sync_tables()
        s = F_.GatherConcat2.apply(self.source_user_embedding.weight, self.source_item_embedding.weight, user, item)
        t = F_.GatherConcat2.apply(self.target_user_embedding.weight, self.target_item_embedding.weight, user, item)
        if self.mode == 'overlap_users':
            m = F_.overlap_mask(user, self.overlapped_num_users)
        else:
            m = F_.overlap_mask(item, self.overlapped_num_items)
        for l in range(len(self.crossparas)):
            ls, lt = self.source_crossunit_linear[l], self.target_crossunit_linear[l]
            s, t = F_.CrossUnit.apply(s, t, ls.weight, ls.bias, lt.weight, lt.bias, self.crossparas[l].weight, m)
        return s, t

    def state_dict(self, *args, **kwargs):
        self.sync_tables()
        return super().state_dict(*args, **kwargs)

    def source_forward(self, user, item):
        s, _ = self._towers(user, item)
        lin = self.source_outputunit[0]
        return F_.linear(s, lin.weight, lin.bias, B_.ACT_SIGMOID).squeeze()

    def target_forward(self, user, item):
        _, t = self._towers(user, item)
        lin = self.target_outputunit[0]
        return F_.linear(t, lin.weight, lin.bias, B_.ACT_SIGMOID).squeeze()

    def _fused_params(self):
        ps = []
        for l in range(len(self.crossparas)):
            ls, lt = self.source_crossunit_linear[l], self.target_crossunit_linear[l]
            ps += [ls.weight, ls.bias, lt.weight, lt.bias, self.crossparas[l].weight]
        so, to = self.source_outputunit[0], self.target_outputunit[0]
        return ps + [so.weight, so.bias, to.weight, to.bias]

    def graph_key(self):
        return ('CoNet', self.fused_towers, self.row_opt is not None)

    # ---- hooks for a software-pipelined captured step (graph_step.GraphedTrainStep) ---------------------------------------------------
    def prepare_batch(self, interaction):
        """The part of ``calculate_loss`` that needs only the batch's ids and the row update of the PREVIOUS step: the id sort and the
        replay of the rows' postponed Adam updates.  ``calculate_loss`` on the same interaction object then skips it.  Returns False when
        there is nothing to run ahead (no deferred optimizer)."""
        if self.row_opt is None or not self.fused_towers:
            return False
        su, si = interaction[self.SOURCE_USER_ID], interaction[self.SOURCE_ITEM_ID]
        tu, ti = interaction[self.TARGET_USER_ID], interaction[self.TARGET_ITEM_ID]
        self.row_opt.prepare([(su, tu), (si, ti)])
        self.row_opt._prepared = id(interaction)
        return True

    def _id_lists(self, interaction):
        return [(interaction[self.SOURCE_USER_ID], interaction[self.TARGET_USER_ID]), (interaction[self.SOURCE_ITEM_ID], interaction[self.TARGET_ITEM_ID])]

    def sort_batch(self, interaction, slot):
        """``prepare_batch`` in two halves for a step pipelined two batches deep: the id sort of a batch (depends on its ids only) ..."""
        if self.row_opt is None or not self.fused_towers:
            return False
        self.row_opt.sort_ahead(self._id_lists(interaction), slot)
        return True

    def replay_batch(self, interaction, slot):
        """... and the replay of its rows' postponed updates (depends on the row update of the step before it)."""
        self.row_opt.prepare_sorted(slot)
        self.row_opt._prepared = id(interaction)

    def apply_rows_early(self):
        """Right behind ``calculate_loss`` of a step whose loss will be differentiated with a unit upstream gradient: launch the
        tables' row update now (the forward launch already produced its gradient rows)."""
        return self.row_opt is not None and self.row_opt.apply_early()

    def calculate_loss(self, interaction):
        # source_forward(source batch) and target_forward(target batch) both run BOTH towers (conet.py:186-187); every
        # op is row-independent, so the two batches go through the cross units as ONE stack of rows and each output unit
        # reads its own slice -- same numbers per row as two separate passes.
        su, si = interaction[self.SOURCE_USER_ID], interaction[self.SOURCE_ITEM_ID]
        tu, ti = interaction[self.TARGET_USER_ID], interaction[self.TARGET_ITEM_ID]
        n_s = su.numel()
        if self.fused_towers:
            # the whole loss as one autograd node on csrc/cdr_conet.hip (one forward launch, three backward launches); the
            # kernel stacks the source and the target batch itself
            over_users = self.mode == 'overlap_users'
            row_opt = self.row_opt if (self.row_opt is not None and torch.is_grad_enabled()) else None
            if row_opt is not None:
                # the batch's rows replay their postponed Adam updates before they are read (unless prepare_batch ran ahead for this batch)
                if row_opt._prepared != id(interaction):
                    row_opt.prepare([(su, tu), (si, ti)])
                row_opt._prepared = None
            else:
                self.sync_tables()
            loss, self.last_loss_parts = F_.ConetFusedLoss.apply(
                self.source_user_embedding.weight, self.source_item_embedding.weight, self.target_user_embedding.weight,
                self.target_item_embedding.weight, su, si, interaction[self.SOURCE_LABEL], tu, ti, interaction[self.TARGET_LABEL],
                self.overlapped_num_users if over_users else self.overlapped_num_items, over_users, self._dims, row_opt,
                *self._fused_params())
            return loss
        user, item = torch.cat([su.reshape(-1), tu.reshape(-1)]), torch.cat([si.reshape(-1), ti.reshape(-1)])
        s, t = self._towers(user, item)
        ls, lt = self.source_outputunit[0], self.target_outputunit[0]
        p_source = F_.linear(s[:n_s], ls.weight, ls.bias, B_.ACT_SIGMOID).squeeze()
        p_target = F_.linear(t[n_s:], lt.weight, lt.bias, B_.ACT_SIGMOID).squeeze()
        loss = F_.BCEProbLoss.apply(p_source, interaction[self.SOURCE_LABEL]) + \
            F_.BCEProbLoss.apply(p_target, interaction[self.TARGET_LABEL])
        for para in self.crossparas:
            loss = loss + F_.FrobeniusNorm.apply(para.weight)
        return loss

    # ---- scoring: target tower without cross terms ---------------------------------------------------------------
    def _target_tower_tail(self, h, first):
        for l in range(first, len(self.target_crossunit_linear)):
            lin = self.target_crossunit_linear[l]
            h = F_.linear(h, lin.weight, lin.bias, B_.ACT_RELU)
        lin = self.target_outputunit[0]
        return F_.linear(h, lin.weight, lin.bias, B_.ACT_SIGMOID)

    @torch.no_grad()
    def predict(self, interaction):
        self.sync_tables()
        x = F_.GatherConcat2.apply(self.target_user_embedding.weight, self.target_item_embedding.weight,
                                   interaction[self.TARGET_USER_ID], interaction[self.TARGET_ITEM_ID])
        return self._target_tower_tail(x, 0)                        # [B,1]

    @torch.no_grad()
    def full_sort_predict(self, interaction):
        """[U,N].  The first layer is separable, W1 [u ; i] = W1u u + W1i i: the item part P = items x W1i^T is computed
        once per call, each user adds its own W1u u + b1 (one broadcast-add-ReLU pass), layers 2.. run as contractions
        over the N rows -- instead of the reference's Python loop over users with a repeat()ed [N, 2D] input."""
        D = self.latent_dim
        self.sync_tables()
        caching = self._eval_caching()
        few = self.__dict__.get('_eval_few') if caching else None
        uid = interaction[self.TARGET_USER_ID]
        if few is not None and few.takes(uid):
            lin1, lo = self.target_crossunit_linear[0], self.target_outputunit[0]
            tail = list(self.target_crossunit_linear)[1:]
            if not few.fresh(self.target_user_embedding.weight, lin1.weight, lin1.bias, [l.weight for l in tail], [l.bias for l in tail], lo.weight, lo.bias):
                self._drop_eval_cache()                                  # a parameter was re-allocated: rebuild P and the pack below
                few = None
        if few is not None and few.takes(uid):
            return few(uid)                                             # a few users, evaluation mode: ONE launch (see below)
        user_e = F_.gather_rows(self.target_user_embedding.weight, uid)
        items = self.target_item_embedding.weight[:self.target_num_items]
        lin1 = self.target_crossunit_linear[0]
        W1 = lin1.weight                                              # [h1, 2D]
        # P depends on the item table and W1 only: inside a freeze_for_eval() bracket (recbole's evaluate calls full_sort_predict once per
        # user batch -- one user per call at the default eval_batch_size) it is formed once and kept until the bracket closes
        P = self.__dict__.get('_eval_P') if caching else None
        if P is None:
            P = F_.gemm(items, W1[:, D:], trans_b=True)                # [N, h1]  (this method runs under no_grad: no graph is kept with it)
            if caching:
                self.__dict__['_eval_P'] = P
        Q = F_.gemm(user_e, W1[:, :D], trans_b=True, bias=lin1.bias)    # [U, h1]
        tail = list(self.target_crossunit_linear)[1:]
        if (self.__dict__.get('fullsort_fused', True) and tail
                and F_.conet_fullsort_supported(W1.shape[0], [l.weight.shape[0] for l in tail])):
            # every (user, item) pair through layers 2.. and the output unit in ONE launch, activations in registers
            # (csrc/cdr_conet_fullsort.hip) -- no per-user loop, no [N, h] intermediates
            lo = self.target_outputunit[0]
            if caching and '_eval_few' not in self.__dict__:
                # recbole's evaluation enters here once per eval batch -- ONE user at the default eval_batch_size over a large catalogue:
                # from the second call on such a call is one launch with Q formed inside it (cdr_conet_fullsort_users), arguments packed once
                self.__dict__['_eval_few'] = F_.ConetFullsortFewUsers(P, self.target_user_embedding.weight, W1, lin1.bias, D, [l.weight for l in tail],
                                                                      [l.bias for l in tail], lo.weight, lo.bias)
            return F_.conet_fullsort(P, Q, [l.weight for l in tail], [l.bias for l in tail], lo.weight, lo.bias)
        rows = []
        for u in range(user_e.shape[0]):
            h = F_.bcast_add_act(P, Q[u], B_.ACT_RELU)
            rows.append(self._target_tower_tail(h, 1).view(1, -1))
        return torch.cat(rows, dim=0)
