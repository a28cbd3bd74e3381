"""Device-side negative sampler mirroring recbole_cdr/sampler/crossdomain_sampler.py (CrossDomainSourceSampler) and
recbole's target-domain sampler: uniform candidates, rejection against the user's used items, k-major output."""
import numpy as np
import torch

from . import binding as B_


class DeviceNegSampler:
    """``sample_by_user_ids(user_ids, item_ids, num)`` like the reference's samplers, but the draw runs on the GPU.

    domain='source': candidates [1, OI) U [OI+TOI, total_items)   (crossdomain_sampler.py:212-214)
    domain='target': candidates [1, OI+TOI)                        (recbole Sampler over the target dataset's items)
    ``used_pairs``: [n, 2] (user, item) interactions whose items must never be returned for that user."""

    def __init__(self, dataset, domain, used_pairs, device, seed=2022, distribution='uniform'):
        OI, TOI = dataset.num_overlap_item, dataset.num_target_only_item
        total_i, total_u = dataset.num_total_item, dataset.num_total_user
        if domain == 'source':
            self.ranges = (1, OI, OI + TOI, total_i)
        else:
            self.ranges = (1, OI + TOI, 0, 0)
        # the per-user CSR of used items, built where it will live: one sort of (user, item) keys on the device (numpy's
        # np.unique(axis=0) over 12 M pairs took ~5 s per sampler)
        pairs_t = torch.as_tensor(np.asarray(used_pairs, dtype=np.int64) if not torch.is_tensor(used_pairs) else used_pairs).to(device)
        key = torch.unique(pairs_t[:, 0] * total_i + pairs_t[:, 1])               # sorted by (user, item), duplicates dropped
        users_sorted = key // total_i
        indptr = torch.zeros(total_u + 1, device=device, dtype=torch.int64)
        indptr[1:] = torch.cumsum(torch.bincount(users_sorted, minlength=total_u), 0)
        n_cand = max(self.ranges[1] - self.ranges[0], 0) + max(self.ranges[3] - self.ranges[2], 0)
        if bool(((indptr[1:] - indptr[:-1]) >= n_cand).any()):
            raise ValueError('Some users have interacted with all items, which we can not sample negative items for them. '
                             'Please set `user_inter_num_interval` to filter those users.')
        self.indptr = indptr
        self.indices = (key % total_i).contiguous()
        self.fail = torch.zeros(1, device=device, dtype=torch.int32)
        self.seed, self.calls, self.device = int(seed), 0, device
        if distribution not in ('uniform', 'popularity'):
            raise NotImplementedError(f'The sampling distribution [{distribution}] is not implemented.')
        self.distribution = distribution
        if distribution == 'popularity':
            keys, prob, alias = build_alias_table(np.asarray(used_pairs, dtype=np.int64)[:, 1])
            self.keys = torch.from_numpy(keys).to(device)
            self.prob = torch.from_numpy(prob.astype(np.float32)).to(device)
            self.alias = torch.from_numpy(alias).to(device)

    def sample_by_user_ids(self, user_ids, item_ids, num):
        users = user_ids.to(self.device).contiguous().to(torch.int64)
        S = users.numel()
        out = torch.empty(S * num, device=self.device, dtype=torch.int64)
        self.calls += 1
        seed = (self.seed * 0x9E3779B1 + self.calls * 0x85EBCA77) & 0xFFFFFFFFFFFFFFFF
        if self.distribution == 'popularity':
            B_.call('cdr_neg_sample_alias', B_.stream(), B_.i64(users), S, int(num), B_.i64(self.keys), B_.f32(self.prob),
                    B_.i64(self.alias), self.keys.numel(), B_.i64(self.indptr), B_.i64(self.indices), seed, B_.i64(out),
                    B_.raw(self.fail))
            return out
        lo0, hi0, lo1, hi1 = self.ranges
        B_.call('cdr_neg_sample_uniform', B_.stream(), B_.i64(users), S, int(num), lo0, hi0, lo1, hi1, B_.i64(self.indptr),
                B_.i64(self.indices), (self.seed * 0x9E3779B1 + self.calls * 0x85EBCA77) & 0xFFFFFFFFFFFFFFFF, B_.i64(out),
                B_.raw(self.fail))
        return out

    __call__ = sample_by_user_ids

    def graph_seed(self):
        """Base seed of the draws made inside a captured batch producer (data/producer.py): the kernel adds its own device-side draw
        count, so hipGraph replays never repeat a draw; a different stream of numbers from ``sample_by_user_ids``' host-counted one."""
        return (self.seed * 0x9E3779B1 + 0x632BE59BD9B4E019) & 0xFFFFFFFFFFFFFFFF

    def check_failures(self):
        """After 64 rejected draws the kernels stop drawing and pick a free candidate directly (uniform sampler: a uniform draw over
        the user's FREE candidates, which is what the reference's redraw loop converges to; popularity sampler: the first unused key
        from a random column on), so a returned negative is never an interacted item.  The device flag is raised only when a user
        has NO free candidate at all -- which the constructor refuses for the uniform ranges, and which can still happen for the
        popularity table (a user who interacted with every item that has any interaction).  One host sync: the loaders call it
        once per epoch, in every mode."""
        if int(self.fail.item()):
            self.fail.zero_()
            raise RuntimeError('DeviceNegSampler: a user has interacted with every candidate item; no negative exists for them. '
                               'Filter such users (`user_inter_num_interval`) as the reference requires.')


def build_alias_table(candidates):
    """Walker alias table over the item ids of the sampler's interactions, as crossdomain_sampler.py:66-94 builds it
    (probabilities scaled to mean 1, large / small queues served first-in first-out).  -> (keys [n] int64 in first-occurrence
    order like ``Counter``, prob [n] float64, alias [n] int64 item ids; an unassigned alias stays -1 and is never drawn
    because its prob is 1)."""
    from collections import Counter, deque
    cnt = Counter(np.asarray(candidates).tolist())
    keys = np.fromiter(cnt.keys(), dtype=np.int64, count=len(cnt))
    prob = np.fromiter(cnt.values(), dtype=np.float64, count=len(cnt)) / len(candidates) * len(cnt)
    alias = np.full(len(cnt), -1, dtype=np.int64)
    large = deque(np.nonzero(prob > 1)[0].tolist())
    small = deque(np.nonzero(prob < 1)[0].tolist())
    while large and small:
        l, s = large.popleft(), small.popleft()
        alias[s] = keys[l]
        prob[l] -= 1 - prob[s]
        if prob[l] < 1:
            small.append(l)
        elif prob[l] > 1:
            large.append(l)
    return keys, prob, alias
