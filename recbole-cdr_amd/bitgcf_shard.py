"""BiTGCF with the embedding tables AND the graph row-sharded over the GPUs of a node (BASELINE configs[3]: "BiTGCF Douban-Book ->
Music 2-layer GCN, item table row-sharded across 4 x MI355X"; SURVEY 8e: E and the CSR sharded by destination row, per-layer
all-gather of E).  Reference math: recbole_cdr/model/cross_domain_recommender/bitgcf.py:130-135 (graph layer), :137-172 (transfer
layer), :174-205 (forward), :207-250 (loss).

Partition.  Users and items are cut into G contiguous blocks each (padded to equal size); rank r owns user block r and item block
r of all four tables, their Adam state, the rows of both normalised adjacencies that belong to those nodes, and the degree
vectors of the transfer layer.  An entity has the same id in both domains, so a rank owns BOTH domains' rows of every node it
owns: the transfer layer (which mixes source row r with target row r) is local.  Everything in the propagation is row-wise
EXCEPT the SpMM, whose gather side needs every row:

    forward, per layer and domain     all-gather E_l (n x D)              -> local SpMM rows + fused layer math
    backward, per layer and domain    all-gather g (.) (1 + E_l)          -> local SpMM rows   (A is symmetric: A^T g = A g)
    after the propagation             the batch's rows to the rank that scores them (below)

The batch loss (``batch_loss='routed'``, what 'auto' picks for more than one rank).  Every rank holds the whole batch's ids (the loader is replicated), so no id
exchange is needed: rank r scores rows [r B/G, (r+1) B/G) of the batch.  Each rank gathers, for EVERY batch row, the user / item row
of the stacked output it owns (zeros otherwise: ``cdr_gather_block_rows``); one reduce-scatter (sum; exactly one addend per row is
non-zero, so the sum is exact) hands every rank the rows of its slice -- 2 B (L+1) D floats on the wire instead of the all-gather of
the whole stacked table (n (L+1) D: 61 MB per domain at Douban sizes against 12.6 MB).  The slice's BCE sum and the two EmbLoss
sums of squares are all-reduced (six floats for both domains), each rank forms the gradient rows of its slice with the GLOBAL
batch size and norms, an all-gather returns them and every rank scatter-adds the rows it owns (``cdr_scatter_add_block_rows``).
Per rank the batch costs 1/G of the scoring and no dense [n, W] gradient of the gathered table.  ``batch_loss='replicated'`` keeps the
round-3 form: all-gather of the stacked outputs, every rank evaluates the whole batch on the gathered tables and keeps its own
segment of the (identical) gradient -- all-gathers only, no reduction.

``ops`` supplies the local arithmetic: ``NativeGraphOps`` (libcdrhip) here, the oracle's torch formulas in the gloo CPU tests.
"""
import numpy as np
import torch
import torch.distributed as dist


class BlockPartition:
    def __init__(self, n_users, n_items, world):
        self.nu, self.ni, self.G = int(n_users), int(n_items), int(world)
        self.bu = -(-self.nu // self.G)
        self.bi = -(-self.ni // self.G)
        self.nl = self.bu + self.bi                      # rows per rank: [user block ; item block]

    def user_pos(self, u):
        """Position of user id u in the all-gathered buffer [G * nl, .]."""
        return (u // self.bu) * self.nl + (u % self.bu)

    def item_pos(self, i):
        return (i // self.bi) * self.nl + self.bu + (i % self.bi)

    def overlap_local(self, n_overlap, rank, block):
        return int(min(max(n_overlap - rank * block, 0), block))

    def shard(self, full, rank, users):
        """Rank's block of a full [rows, ...] table, zero-padded to the block size."""
        b = self.bu if users else self.bi
        lo = rank * b
        part = full[lo:lo + b]
        if part.shape[0] < b:
            pad = torch.zeros((b - part.shape[0],) + tuple(full.shape[1:]), dtype=full.dtype, device=full.device)
            part = torch.cat([part, pad], dim=0)
        return part.contiguous()

    def unshard(self, gathered, users):
        """[G * nl, W] all-gathered buffer -> the full [nu or ni, W] table in id order."""
        W = gathered.shape[1]
        g = gathered.view(self.G, self.nl, W)
        blk = g[:, :self.bu] if users else g[:, self.bu:]
        return blk.reshape(-1, W)[:self.nu if users else self.ni]


def local_norm_adj(pairs, part, rank):
    """This rank's rows of D^-1/2 A D^-1/2 (values formed exactly as bitgcf.py:103-115: float64 product with degree + 1e-7, rounded to
    fp32), as CSR over the rank's local row order with column indices into the all-gathered buffer."""
    nu, ni = part.nu, part.ni
    pairs = np.unique(np.asarray(pairs, dtype=np.int64), axis=0)
    u, i = pairs[:, 0], pairs[:, 1]
    deg_u = np.bincount(u, minlength=nu).astype(np.float64) + 1e-7
    deg_i = np.bincount(i, minlength=ni).astype(np.float64) + 1e-7
    du, di = np.power(deg_u, -0.5), np.power(deg_i, -0.5)
    upos = (u // part.bu) * part.nl + (u % part.bu)
    ipos = (i // part.bi) * part.nl + part.bu + (i % part.bi)
    mine_u = (u // part.bu) == rank                       # rows of my users: neighbours are items
    mine_i = (i // part.bi) == rank
    rows = np.concatenate([u[mine_u] % part.bu, part.bu + (i[mine_i] % part.bi)])
    cols = np.concatenate([ipos[mine_u], upos[mine_i]])
    vals = np.concatenate([(du[u[mine_u]] * np.float64(1.0)) * di[i[mine_u]], (di[i[mine_i]] * np.float64(1.0)) * du[u[mine_i]]]).astype(np.float32)
    # the single-process CSR orders a row's entries by global column (users first, then items); keep that order so that the fp32
    # row sums add in the same sequence: sort by (row, global column)
    gcol = np.concatenate([nu + i[mine_u], u[mine_i]])
    order = np.lexsort((gcol, rows))
    rows, cols, vals = rows[order], cols[order], vals[order]
    indptr = np.zeros(part.nl + 1, dtype=np.int64)
    np.cumsum(np.bincount(rows, minlength=part.nl), out=indptr[1:])
    return indptr, cols.astype(np.int64), vals


class NativeGraphOps:
    """libcdrhip kernels on device tensors (csrc/cdr_graph.hip, cdr_gather_loss.hip)."""

    def __init__(self, device):
        from . import binding as B_, functional as F_
        self.B_, self.F_, self.device = B_, F_, device

    def tensor(self, a):
        return torch.as_tensor(a).to(self.device)

    def graph_layer_fwd(self, csr, Eg, E):
        B_ = self.B_
        side, new = torch.empty_like(E), torch.empty_like(E)
        B_.call('cdr_graph_layer_fwd_rows', B_.stream(), B_.i64(csr[0]), B_.i64(csr[1]), B_.f32(csr[2]), E.shape[0], B_.f32(Eg), B_.f32(E),
                E.shape[1], B_.f32(side), B_.f32(new))
        return side, new

    def mul_one_plus(self, g, x):
        B_ = self.B_
        out = torch.empty_like(g)
        B_.call('cdr_mul_one_plus', B_.stream(), B_.f32(g), B_.f32(x), g.numel(), B_.f32(out))
        return out

    def graph_layer_bwd(self, csr, tmp_g, gnew, side):
        B_ = self.B_
        gE = torch.empty_like(gnew)
        B_.call('cdr_graph_layer_bwd_rows', B_.stream(), B_.i64(csr[0]), B_.i64(csr[1]), B_.f32(csr[2]), gnew.shape[0], B_.f32(tmp_g),
                B_.f32(gnew), B_.f32(side), gnew.shape[1], B_.f32(gE))
        return gE

    def transfer_fwd(self, S, T, ds, dt, n_overlap, lam_s, lam_t):
        B_ = self.B_
        So, To = torch.empty_like(S), torch.empty_like(T)
        B_.call('cdr_transfer_fwd', B_.stream(), B_.f32(S), B_.f32(T), B_.f32(ds), B_.f32(dt), S.shape[0], S.shape[1], int(n_overlap),
                float(lam_s), float(lam_t), B_.f32(So), B_.f32(To))
        return So, To

    def transfer_bwd(self, gSo, gTo, ds, dt, n_overlap, lam_s, lam_t):
        B_ = self.B_
        gS, gT = torch.empty_like(gSo), torch.empty_like(gTo)
        B_.call('cdr_transfer_bwd', B_.stream(), B_.f32(gSo), B_.f32(gTo), B_.f32(ds), B_.f32(dt), gSo.shape[0], gSo.shape[1], int(n_overlap),
                float(lam_s), float(lam_t), B_.f32(gS), B_.f32(gT))
        return gS, gT

    def dropout(self, x, p, seed):
        B_ = self.B_
        out = torch.empty_like(x)
        B_.call('cdr_dropout', B_.stream(), B_.f32(x), x.numel(), float(p), int(seed), B_.f32(out))
        return out

    def l2norm_fwd(self, x):
        B_ = self.B_
        y, nrm = torch.empty_like(x), torch.empty(x.shape[0], device=x.device, dtype=torch.float32)
        B_.call('cdr_l2_normalize_fwd', B_.stream(), B_.f32(x), x.shape[0], x.shape[1], B_.f32(y), x.shape[1], B_.f32(nrm))
        return y, nrm

    def l2norm_bwd(self, x, nrm, gy):
        B_ = self.B_
        gx = torch.empty_like(x)
        B_.call('cdr_l2_normalize_bwd', B_.stream(), B_.f32(x), B_.f32(nrm), B_.f32(gy.contiguous()), x.shape[1], x.shape[0], x.shape[1],
                B_.f32(gx), 0)
        return gx

    # ---- routed batch loss: rows of this rank's slice of the batch -------------------------------------------------------------------
    def gather_owned(self, X, pos, lo):
        """out[r] = X[pos[r] - lo] where this rank owns position pos[r] of the gathered layout, else 0 (pos < 0: padding)."""
        B_ = self.B_
        out = torch.empty(pos.numel(), X.shape[1], device=X.device, dtype=torch.float32)
        B_.call('cdr_gather_block_rows', B_.stream(), B_.f32(X), X.shape[1], B_.i64(pos), pos.numel(), int(lo), X.shape[0], B_.f32(out))
        return out

    def scatter_owned(self, rows, W, pos, lo, src):
        B_ = self.B_
        g = torch.zeros(rows, W, device=src.device, dtype=torch.float32)
        B_.call('cdr_scatter_add_block_rows', B_.stream(), B_.f32(g), W, B_.i64(pos), pos.numel(), int(lo), rows, B_.f32(src.contiguous()))
        return g

    def slice_partials(self, rows, Bs, n, W, D, concat, label):
        """This rank's slice of the batch as delivered by the reduce-scatter: rows [2 Bs, Wx] = [user rows ; item rows] (the first n of each
        half are real), Wx = W (concat: the ego rows are the first D columns) or W + D (mean: ego rows behind the W output columns).
        -> (state, [sum of BCE terms, sum ||ego user rows||^2, sum ||ego item rows||^2]): the three numbers that are all-reduced."""
        F_, B_ = self.F_, self.B_
        dev = rows.device
        if n == 0:
            return None, torch.zeros(3, device=dev, dtype=torch.float32)
        key = (n, Bs, str(dev))
        if getattr(self, '_ar_key', None) != key:
            ar = torch.arange(n, device=dev, dtype=torch.int64)
            self._ar_key, self._ar = key, (ar, ar + Bs)
        au, ai = self._ar
        a = (rows.detach() if rows.shape[1] == W else rows[:, :W].contiguous()).requires_grad_(True)      # the compact [2 Bs, W] "table"
        bce, _ = F_.PointGatherLoss.apply(B_.CDR_LOSS_BCE, a, a, None, None, au, ai, label, 0.0)         # mean over the n rows
        e = (rows[:, :D] if concat else rows[:, W:]).contiguous()
        out3 = torch.empty(3, device=dev, dtype=torch.float32)
        B_.call('cdr_embloss_fwd', B_.ctx(dev), B_.stream(), B_.f32(e), B_.f32(e), D, B_.i64(au), B_.i64(ai), n, B_.f32(out3))
        part = torch.cat([bce.detach().reshape(1) * float(n), out3[1:3] * out3[1:3]])
        return (a, bce, e, n, concat, D, au, ai), part

    def slice_grads(self, state, totals, B, reg_weight):
        """Gradient rows [2 Bs, Wx] of the slice for the GLOBAL batch: d/d rows of [BCE-sum / B + reg_weight (||Eu|| + ||Ei||) / B] with the
        all-reduced sums (recbole's EmbLoss: un-squared norms of the whole batch's rows)."""
        B_ = self.B_
        a, bce, e, n, concat, D, au, ai = state
        (bce.sum() * (float(n) / float(B))).backward()
        out3 = torch.cat([totals[:1], torch.sqrt(totals[1:3])]).contiguous()          # [., ||Eu||, ||Ei||] of the whole batch
        go = torch.full((1,), float(reg_weight) * float(n) / float(B), device=e.device, dtype=torch.float32)
        ge = torch.zeros_like(e)
        B_.call('cdr_embloss_bwd_dense', B_.stream(), B_.f32(e), B_.f32(e), D, B_.i64(au), B_.i64(ai), n, B_.f32(out3), B_.f32(go),
                B_.f32(ge), B_.f32(ge))
        if concat:
            g = a.grad
            g[:, :D] += ge
            return g
        return torch.cat([a.grad, ge], 1)

    def batch_loss(self, out_g, E0_g, pu, pi, label, reg_weight):
        """bitgcf.py:221-247 on the all-gathered tables (ids already mapped to gathered positions): BCE(sigmoid(<u, i>)) +
        reg_weight * EmbLoss(ego rows).  Returns (loss, d loss / d out_g, d loss / d E0_g), dense."""
        F_, B_ = self.F_, self.B_
        a = out_g.detach().requires_grad_(True)
        e = E0_g.detach().requires_grad_(True)
        bce, _ = F_.PointGatherLoss.apply(B_.CDR_LOSS_BCE, a, a, None, None, pu, pi, label, 0.0)
        reg = F_.EmbLossRows.apply(e, e, pu, pi)
        loss = bce + reg_weight * reg
        loss.sum().backward()
        return loss.detach().reshape(()), a.grad, e.grad


class ShardedBiTGCF:
    """One rank's share of BiTGCF.  ``init``: dict of the four FULL tables (reference naming) every rank slices its blocks from
    (tests / small graphs), or None for a rank-local xavier draw."""

    TABLES = ('source_user_embedding.weight', 'source_item_embedding.weight', 'target_user_embedding.weight',
              'target_item_embedding.weight')

    def __init__(self, n_users, n_items, n_overlap_users, n_overlap_items, s_pairs, t_pairs, embedding_size, n_layers, lambda_source,
                 lambda_target, connect_way, reg_weight, ops, group=None, init=None, seed=2022, drop_rate=0.0, batch_loss='auto'):
        assert batch_loss in ('auto', 'routed', 'replicated'), batch_loss
        self.group = group
        self.G = dist.get_world_size(group) if dist.is_initialized() else 1
        # 'auto': routed rows wherever there is a wire to save; one rank alone has none, and the routed form's ~27 extra small launches
        # cost it 0.25 ms per step there (profiles/r04_c4_batch_loss.txt)
        self.batch_loss = ('routed' if self.G > 1 else 'replicated') if batch_loss == 'auto' else batch_loss
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.part = BlockPartition(n_users, n_items, self.G)
        self.ops, self.D, self.L = ops, int(embedding_size), int(n_layers)
        self.lam_s, self.lam_t, self.connect_way, self.reg_weight = float(lambda_source), float(lambda_target), connect_way, float(reg_weight)
        self.drop_rate, self.training = float(drop_rate), True     # nn.Dropout on every graph layer's output (bitgcf.py:66,134)
        p, r = self.part, self.rank
        self.OU_l, self.OI_l = p.overlap_local(n_overlap_users, r, p.bu), p.overlap_local(n_overlap_items, r, p.bi)
        T = ops.tensor
        self.csr = {}
        for dom, pairs in (('s', s_pairs), ('t', t_pairs)):
            ip, ix, vv = local_norm_adj(pairs, p, r)
            self.csr[dom] = (T(ip), T(ix), T(vv))
        deg = lambda pairs, axis, n: np.bincount(np.asarray(pairs)[:, axis], minlength=n).astype(np.float32)
        full_deg = {'su': deg(s_pairs, 0, p.nu), 'tu': deg(t_pairs, 0, p.nu), 'si': deg(s_pairs, 1, p.ni), 'ti': deg(t_pairs, 1, p.ni)}
        self.deg = {k: T(p.shard(torch.from_numpy(v), r, users=k.endswith('u'))) for k, v in full_deg.items()}
        self.params = {}
        gen = torch.Generator().manual_seed(seed + 7919 * r)
        for name in self.TABLES:
            users = '_user_' in name
            if init is not None:
                blk = p.shard(init[name].detach().to('cpu', torch.float32), r, users)
            else:
                rows = p.nu if users else p.ni
                blk = torch.randn(p.bu if users else p.bi, self.D, generator=gen) * (2.0 / (rows + self.D)) ** 0.5
                lo = r * (p.bu if users else p.bi)
                blk[max(rows - lo, 0):] = 0.0                 # padding rows beyond the table stay zero
            self.params[name] = torch.nn.Parameter(T(blk))

    # ---- collectives -------------------------------------------------------------------------------------------------
    def gather(self, x):
        return self.gather_finish(self.gather_start(x))

    def gather_start(self, x):
        """Issue the all-gather of this rank's rows and return a handle: the two domains of a layer are independent until the
        transfer, so both gathers are issued before the first SpMM and the second one travels under it (``gather_finish``)."""
        if self.G == 1:
            return x, None
        x = x.contiguous()
        out = torch.empty((self.G * x.shape[0],) + tuple(x.shape[1:]), device=x.device, dtype=x.dtype)
        return out, (dist.all_gather_into_tensor(out, x, group=self.group, async_op=True), x)     # x kept alive until the wait

    def gather_finish(self, handle):
        out, work = handle
        if work is not None:
            work[0].wait()                      # stream-level wait on the GPU: the host does not block
        return out

    def _is_gloo(self):
        return self.G > 1 and dist.get_backend(self.group) == 'gloo'          # the functional-test transport has no reduce_scatter

    def reduce_scatter_rows(self, S):
        """S [G, m, W]: chunk r summed over the ranks -> [m, W] on rank r."""
        if self.G == 1:
            return S[0]
        if self._is_gloo():
            dist.all_reduce(S, group=self.group)
            return S[self.rank].clone()
        out = torch.empty(tuple(S.shape[1:]), device=S.device, dtype=S.dtype)
        dist.reduce_scatter_tensor(out, S, group=self.group)
        return out

    def all_gather_rows(self, x):
        """[m, W] of every rank -> [G * m, W] in rank order."""
        if self.G == 1:
            return x
        x = x.contiguous()
        out = torch.empty((self.G * x.shape[0],) + tuple(x.shape[1:]), device=x.device, dtype=x.dtype)
        dist.all_gather_into_tensor(out, x, group=self.group)
        return out

    # ---- propagation ---------------------------------------------------------------------------------------------------
    def _propagate(self, want_ego_gathered=True):
        p, ops, D = self.part, self.ops, self.D
        P = self.params
        E = {'s': torch.cat([P[self.TABLES[0]].data, P[self.TABLES[1]].data], 0), 't': torch.cat([P[self.TABLES[2]].data, P[self.TABLES[3]].data], 0)}
        stack = {'s': [E['s']], 't': [E['t']]}
        saved = []
        E0_g = None
        seeds = None
        if self.training and self.drop_rate > 0.0:
            seeds = int(torch.empty((), dtype=torch.int64).random_(0, 2 ** 40).item()) + (self.rank << 44)
        self._seeds = seeds
        for _ in range(self.L):
            pend = {d: self.gather_start(E[d]) for d in 'st'}       # both in flight; 't' arrives under the SpMM of 's'
            Eg, side, new = {}, {}, {}
            for k, d in enumerate('st'):
                Eg[d] = self.gather_finish(pend[d])
                side[d], new[d] = ops.graph_layer_fwd(self.csr[d], Eg[d], E[d])
                if seeds is not None:                       # one counter-based mask per (layer, domain, rank); re-made in the backward
                    new[d] = ops.dropout(new[d], self.drop_rate, seeds + 2 * len(saved) + k)
            if E0_g is None:
                E0_g = Eg
            Su, Tu = ops.transfer_fwd(new['s'][:p.bu], new['t'][:p.bu], self.deg['su'], self.deg['tu'], self.OU_l, self.lam_s, self.lam_t)
            Si, Ti = ops.transfer_fwd(new['s'][p.bu:], new['t'][p.bu:], self.deg['si'], self.deg['ti'], self.OI_l, self.lam_s, self.lam_t)
            S2 = {'s': torch.cat([Su, Si], 0), 't': torch.cat([Tu, Ti], 0)}
            nrm = {}
            for d in 'st':
                y, nrm[d] = ops.l2norm_fwd(S2[d])
                stack[d].append(y)
            saved.append((E, side, S2, nrm))
            E = S2
        if E0_g is None and want_ego_gathered:
            E0_g = {d: self.gather(stack[d][0]) for d in 'st'}
        nb = self.L + 1
        if self.connect_way == 'concat':
            out = {d: torch.cat(stack[d], 1) for d in 'st'}
        else:
            out = {d: torch.stack(stack[d], 1).mean(1) for d in 'st'}
        return out, E0_g, saved, nb, {d: stack[d][0] for d in 'st'}

    def _routed_batch_loss(self, inter, out, ego):
        """The batch loss with rank r scoring rows [r Bs, (r+1) Bs) of the (replicated) batch: rows by reduce-scatter, three sums per domain by
        all-reduce, gradient rows by all-gather (module docstring).  -> (losses, g_out, g_E0) with g_* = this rank's [nl, .] segments."""
        p, ops, D, G, r = self.part, self.ops, self.D, self.G, self.rank
        lo = r * p.nl
        concat = self.connect_way == 'concat'                    # then the ego rows are the first D columns of the stacked output
        st, parts = {}, []
        for d, pre in (('s', 'source'), ('t', 'target')):
            uid, iid = inter[f'{pre}_user_id'].reshape(-1), inter[f'{pre}_item_id'].reshape(-1)
            B = uid.numel()
            Bs = -(-B // G)
            pad = G * Bs - B
            pu, pi = p.user_pos(uid), p.item_pos(iid)
            if pad:
                fill = torch.full((pad,), -1, device=pu.device, dtype=pu.dtype)
                pu, pi = torch.cat([pu, fill]), torch.cat([pi, fill])
            posS = torch.stack([pu.view(G, Bs), pi.view(G, Bs)], 1).reshape(-1).contiguous()      # [G][user rows of slice g ; item rows of slice g]
            X = out[d] if concat else torch.cat([out[d], ego[d]], 1)
            W = out[d].shape[1]
            rows = self.reduce_scatter_rows(ops.gather_owned(X.contiguous(), posS, lo).view(G, 2 * Bs, X.shape[1]))
            n = max(0, min(Bs, B - r * Bs))                                                       # real rows of my slice
            label = inter[f'{pre}_label'].reshape(-1).float()[r * Bs:r * Bs + n].contiguous()
            state, part = ops.slice_partials(rows, Bs, n, W, D, concat, label)
            st[d] = (state, posS, B, Bs, n, X.shape[1], W)
            parts.append(part)
        totals = torch.cat(parts)
        if G > 1:
            dist.all_reduce(totals, group=self.group)                                             # six floats: both domains at once
        t2 = totals.view(2, 3)
        bkey = (st['s'][2], st['t'][2])
        if getattr(self, '_bd_key', None) != bkey:               # (a host-to-device copy per step would drain the stream: cached)
            self._bd_key, self._bd = bkey, torch.tensor([float(bkey[0]), float(bkey[1])], device=totals.device)
        Bd = self._bd
        lvec = t2[:, 0] / Bd + self.reg_weight * (torch.sqrt(t2[:, 1]) + torch.sqrt(t2[:, 2])) / Bd
        losses, g_out, g_E0 = [lvec[0], lvec[1]], {}, {}
        for k, d in enumerate('st'):
            state, posS, B, Bs, n, Wx, W = st[d]
            grow = ops.slice_grads(state, totals[3 * k:3 * k + 3], B, self.reg_weight) if n else \
                torch.zeros(2 * Bs, Wx, device=totals.device, dtype=torch.float32)
            g = ops.scatter_owned(p.nl, Wx, posS, lo, self.all_gather_rows(grow))
            g_out[d] = g[:, :W]
            g_E0[d] = None if concat else g[:, W:]
        return losses, g_out, g_E0

    def loss_and_grads(self, inter):
        """(loss_source, loss_target) of the FULL batch ``inter`` (every rank passes the same batch) and this rank's gradient blocks in
        ``self.params[...].grad``."""
        p, ops, D = self.part, self.ops, self.D
        out, E0_g, saved, nb, ego = self._propagate(want_ego_gathered=self.batch_loss == 'replicated')
        losses, g_out, g_E0 = [], {}, {}
        if self.batch_loss == 'routed':
            losses, g_out, g_E0 = self._routed_batch_loss(inter, out, ego)
        else:
            seg = slice(self.rank * p.nl, (self.rank + 1) * p.nl)
            pend = {d: self.gather_start(out[d]) for d in 'st'}         # the target stack arrives under the source batch loss
            for d, pre in (('s', 'source'), ('t', 'target')):
                out_g = self.gather_finish(pend[d])
                pu, pi = p.user_pos(inter[f'{pre}_user_id'].reshape(-1)), p.item_pos(inter[f'{pre}_item_id'].reshape(-1))
                loss, go, ge = ops.batch_loss(out_g, E0_g[d], pu, pi, inter[f'{pre}_label'].reshape(-1).float(), self.reg_weight)
                losses.append(loss)
                g_out[d], g_E0[d] = go[seg], ge[seg]                # identical on every rank: keep the own segment
        # ---- backward through the stack, last layer first
        if self.connect_way == 'concat':
            g_blk = {d: [g_out[d][:, b * D:(b + 1) * D] for b in range(nb)] for d in 'st'}
        else:
            g_blk = {d: [g_out[d] / nb for _ in range(nb)] for d in 'st'}
        g_next = {d: None for d in 'st'}
        for l in reversed(range(self.L)):
            E, side, S2, nrm = saved[l]
            gS2 = {}
            for d in 'st':
                g = ops.l2norm_bwd(S2[d], nrm[d], g_blk[d][l + 1])
                gS2[d] = g if g_next[d] is None else g + g_next[d]
            gSu, gTu = ops.transfer_bwd(gS2['s'][:p.bu], gS2['t'][:p.bu], self.deg['su'], self.deg['tu'], self.OU_l, self.lam_s, self.lam_t)
            gSi, gTi = ops.transfer_bwd(gS2['s'][p.bu:], gS2['t'][p.bu:], self.deg['si'], self.deg['ti'], self.OI_l, self.lam_s, self.lam_t)
            gnew = {'s': torch.cat([gSu, gSi], 0), 't': torch.cat([gTu, gTi], 0)}
            if self._seeds is not None:
                gnew = {d: ops.dropout(gnew[d], self.drop_rate, self._seeds + 2 * l + k) for k, d in enumerate('st')}
            pend = {d: self.gather_start(ops.mul_one_plus(gnew[d], E[d])) for d in 'st'}   # A is symmetric: A^T g = A g on the gathered g (1 + E)
            for d in 'st':
                g_next[d] = ops.graph_layer_bwd(self.csr[d], self.gather_finish(pend[d]), gnew[d], side[d])
        for d, names in (('s', self.TABLES[:2]), ('t', self.TABLES[2:])):
            g = g_blk[d][0] if g_E0[d] is None else g_blk[d][0] + g_E0[d]
            if g_next[d] is not None:
                g = g + g_next[d]
            self.params[names[0]].grad = g[:p.bu].contiguous()
            self.params[names[1]].grad = g[p.bu:].contiguous()
        return tuple(losses)

    @torch.no_grad()
    def propagated_tables(self):
        """The four FULL propagated tables (evaluation: bitgcf.py:264-282; dropout off), assembled from every rank's rows."""
        was, self.training = self.training, False
        try:
            out, _, _, _, _ = self._propagate()
        finally:
            self.training = was
        res = []
        for d in 'st':
            g = self.gather(out[d])
            res += [self.part.unshard(g, True), self.part.unshard(g, False)]
        return res

    @torch.no_grad()
    def full_tables(self):
        """The four FULL embedding tables in id order (tests, checkpoints)."""
        res = {}
        for d, names in (('s', self.TABLES[:2]), ('t', self.TABLES[2:])):
            g = self.gather(torch.cat([self.params[names[0]].data, self.params[names[1]].data], 0))
            res[names[0]], res[names[1]] = self.part.unshard(g, True), self.part.unshard(g, False)
        return res
