"""Data parallelism for the drop-in models (BASELINE C2-C4: EMCDR, CMF, CoNet, SSCDR, BiTGCF) with the parameters' optimizer
state row-sharded over the GPUs of a node -- "item table row-sharded across 4 x MI355X" (BASELINE configs[3]) for tables
that are small enough to replicate for compute but whose Adam sweep (the dominant cost of those steps: 7 x 4 B per element
per step) need not be repeated on every GPU.

One flat fp32 buffer holds every parameter (the model's tensors become views of it); rank r owns the contiguous slice
``[r*chunk, (r+1)*chunk)`` of it together with that slice's exp_avg / exp_avg_sq.  A step is
    local calculate_loss + backward on this rank's batch                       (the single-GPU kernels, unchanged)
 -> ONE reduce-scatter of the flat gradient (mean over ranks)                  (RCCL; 4 B per parameter element)
 -> exact dense Adam on the owned slice only (cdr_adam_multi_dev; 1/G of the sweep), per-PARAMETER step counts as in
    torch.optim.Adam: a parameter that got no gradient in this phase keeps its moments and its count
 -> ONE all-gather of the updated slices back into the flat buffer            (in place).
Semantics = torch DistributedDataParallel + Adam: the update uses the MEAN over ranks of the per-rank gradients (each rank's
loss is the reference's loss on its own batch).  The reference has no multi-GPU code; parity is against that definition
evaluated in one process (tests/test_gpu_parity.py::test_sharded_data_parallel_*).
"""
import ctypes

import torch
import torch.distributed as dist

from . import binding as B_


class ShardedDataParallel:
    def __init__(self, model, group=None, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, adam_impl=None):
        self.model, self.group = model, group
        self.adam_impl = adam_impl       # CPU (gloo) tests inject stand-in arithmetic; the product default is libcdrhip only
        self.world, self.rank = dist.get_world_size(group), dist.get_rank(group)
        self.lr, self.betas, self.eps, self.wd = lr, betas, eps, weight_decay
        self.params = [p for p in model.parameters() if p.requires_grad]
        dev = self.params[0].device
        self.offsets, total = [], 0
        for p in self.params:
            self.offsets.append(total)
            total += p.numel()
        gran = 64 * self.world                                       # 256-byte aligned slices
        self.total = (total + gran - 1) // gran * gran
        self.chunk = self.total // self.world
        self.lo, self.hi = self.rank * self.chunk, (self.rank + 1) * self.chunk
        self.flat = torch.zeros(self.total, device=dev, dtype=torch.float32)
        for p, o in zip(self.params, self.offsets):
            self.flat[o:o + p.numel()].copy_(p.data.reshape(-1))
            p.data = self.flat[o:o + p.numel()].view(p.shape)        # the model now computes straight from the flat buffer
        dist.broadcast(self.flat, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
        self.gflat = torch.zeros(self.total, device=dev, dtype=torch.float32)
        self.gshard = torch.zeros(self.chunk, device=dev, dtype=torch.float32)
        self.pshard = self.flat[self.lo:self.hi]                     # view: Adam updates the flat buffer in place
        self.exp_avg = torch.zeros(self.chunk, device=dev, dtype=torch.float32)
        self.exp_avg_sq = torch.zeros(self.chunk, device=dev, dtype=torch.float32)
        self.steps = torch.zeros(len(self.params), device=dev, dtype=torch.int64)      # device-resident, one per parameter
        self._gloo = dist.get_backend(group) == 'gloo'               # functional-test transport: no reduce_scatter there

    def step(self, interaction):
        """One training step on this rank's batch; returns this rank's loss (detached)."""
        for p in self.params:
            p.grad = None
        losses = self.model.calculate_loss(interaction)
        loss = (sum(losses) if isinstance(losses, tuple) else losses).sum()
        loss.backward()
        self.gflat.zero_()
        touched = []
        for i, (p, o) in enumerate(zip(self.params, self.offsets)):
            if p.grad is not None:                                   # same set on every rank (same model, same phase)
                self.gflat[o:o + p.numel()].copy_(p.grad.reshape(-1))
                touched.append(i)
        if self._gloo:
            dist.all_reduce(self.gflat, group=self.group)
            self.gshard.copy_(self.gflat[self.lo:self.hi])
        else:
            dist.reduce_scatter_tensor(self.gshard, self.gflat, group=self.group)
        self.gshard.mul_(1.0 / self.world)
        segs = []
        for i in touched:
            a, b = max(self.offsets[i], self.lo), min(self.offsets[i] + self.params[i].numel(), self.hi)
            if a < b:
                segs.append((a - self.lo, b - a, i))
        if segs and self.adam_impl is not None:
            self.adam_impl(self, segs)
        elif segs:
            n = len(segs)
            arr = lambda xs: (ctypes.c_void_p * n)(*[x.value for x in xs])
            sl = lambda t: [B_.f32(t[a:a + m]) for a, m, _ in segs]
            B_.call('cdr_adam_multi_dev', B_.stream(), n, arr(sl(self.pshard)), arr(sl(self.gshard)), arr(sl(self.exp_avg)),
                    arr(sl(self.exp_avg_sq)), (ctypes.c_int64 * n)(*[m for _, m, _ in segs]),
                    arr([B_.i64(self.steps[i:i + 1]) for _, _, i in segs]), float(self.lr), float(self.betas[0]),
                    float(self.betas[1]), float(self.eps), float(self.wd), None, None, None)
        if self._gloo:
            dist.all_gather_into_tensor(self.flat, self.pshard.clone(), group=self.group)
        else:
            dist.all_gather_into_tensor(self.flat, self.pshard, group=self.group)         # in place: slice r of the output
        return loss.detach()
