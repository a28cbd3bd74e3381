"""hipGraph capture of one drop-in training step (calculate_loss -> backward -> dense Adam).

On small, L2-resident configurations (BASELINE C1-C4) the reference's step is hundreds of tiny launches; ours is fewer
but still launch/host bound from Python.  Every native call enqueues on torch's current stream and allocates through
torch's caching allocator, so the whole step can be captured once into a hipGraph and replayed: the host then pays one
graph launch per step instead of one ctypes call + one allocation per kernel.  Batch tensors are static buffers that
``step`` copies the new ids / labels into (same shapes every step; a ragged last batch runs eagerly).
"""
import torch


class GraphedTrainStep:
    def __init__(self, model, optimizer, example_interaction, warmup=3, restore_after_warmup=True):
        """``restore_after_warmup``: the warm-up runs REAL steps on ``example_interaction`` (every lazily created buffer, native context
        and optimizer state must exist before the capture); with True the parameters and the optimizer state are put back afterwards,
        so swapping the eager loop for the graphed one does not add ``warmup`` extra updates on batch 0."""
        self.model, self.optimizer = model, optimizer
        self.restore_after_warmup = restore_after_warmup
        # the batch fields live side by side in ONE byte buffer (16-B aligned views): a producer that hands over a batch already
        # packed this way (``pack``) costs one device copy per step instead of one per field (6-7 for a two-domain pointwise batch:
        # 30-40 us, a sixth of the C3 step)
        self._layout, off = {}, 0
        for k, v in example_interaction.items():
            nb = v.numel() * v.element_size()
            self._layout[k] = (off, nb, v.dtype, tuple(v.shape))
            off += (nb + 15) // 16 * 16
        dev = next(iter(example_interaction.values())).device
        self.flat = torch.empty(max(off, 16), device=dev, dtype=torch.uint8)
        self.static = {k: self.flat[o:o + nb].view(dt).view(shape) for k, (o, nb, dt, shape) in self._layout.items()}
        for k, v in example_interaction.items():
            self.static[k].copy_(v)
        self.graph = None
        self.loss = None
        self._one = torch.ones((), device=dev, dtype=torch.float32)       # d loss / d loss, made once: backward() would fill one per step
        self._capture(warmup)

    def _eager(self, interaction):
        # set_to_none: autograd then hands each parameter its gradient buffer instead of "zero-fill, then add into it" (two
        # extra kernels per parameter per step); inside the capture those buffers come from the graph's private pool, so
        # their addresses are the same on every replay
        self.optimizer.zero_grad(set_to_none=True)
        losses = self.model.calculate_loss(interaction)
        loss = sum(losses) if isinstance(losses, tuple) else losses
        if loss.dim():
            loss = loss.reshape(()) if loss.numel() == 1 else loss.sum()   # (a [1]-shaped loss: a view, not a reduction launch)
        loss.backward(self._one if loss.dtype == torch.float32 else None)
        self.optimizer.step()
        return loss.detach()

    def _capture(self, warmup):
        # populate .grad and the optimizer state (and every lazily created native context) before capture
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        snap = self._snapshot() if self.restore_after_warmup else None
        with torch.cuda.stream(side):
            for i in range(max(warmup, 1)):
                self._eager(self.static)
                if i == 0 and snap is not None:
                    snap = self._snapshot(snap)           # optimizer state created by the first step: remembered as zeros
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        if snap is not None:
            self._restore(snap)
        self.graph = torch.cuda.CUDAGraph()
        # capture on the stream that ran the warm-up: the native contexts are per (device, stream) and creating one
        # allocates, which is not allowed while a stream is capturing
        with torch.cuda.graph(self.graph, stream=side):
            self.loss = self._eager(self.static)

    # ---- warm-up without side effects: in-place snapshot / restore (addresses must not change: the capture follows) -------------
    def _state_tensors(self):
        ts = [p.data for p in self.model.parameters()]
        for st in self.optimizer.state.values():
            ts += [v for v in st.values() if torch.is_tensor(v)]
        ro = getattr(self.optimizer, 'row_opt', None)
        if ro is not None:
            ts += ro.exp_avg + ro.exp_avg_sq + ro.last + [ro.counters]
        ds = getattr(self.model, '_drop_state', None)
        if ds is not None:
            ts.append(ds)
        return ts

    def _snapshot(self, prev=None):
        ts = self._state_tensors()
        if prev is None:
            ro = getattr(self.optimizer, 'row_opt', None)
            return {'tensors': [(t, t.clone()) for t in ts], 'row': None if ro is None else (ro.step_count, ro.dirty)}
        known = {t.data_ptr() for t, _ in prev['tensors']}
        prev['tensors'] += [(t, torch.zeros_like(t)) for t in ts if t.data_ptr() not in known]   # state born in step 1 started at zero
        return prev

    def _restore(self, snap):
        with torch.no_grad():
            for t, c in snap['tensors']:
                t.copy_(c)
        ro = getattr(self.optimizer, 'row_opt', None)
        if ro is not None and snap['row'] is not None:
            ro.step_count, ro.dirty = snap['row']

    def pack(self, interaction):
        """The batch in the captured step's own byte layout (a new uint8 tensor): ``step(packed)`` then needs ONE device copy."""
        out = torch.empty_like(self.flat)
        for k, (o, nb, dt, shape) in self._layout.items():
            out[o:o + nb].view(dt).view(shape).copy_(interaction[k])
        return out

    def step(self, interaction):
        if torch.is_tensor(interaction):                      # a batch packed by ``pack``
            assert interaction.dtype == torch.uint8 and interaction.numel() == self.flat.numel(), 'not a batch packed for this step'
            self.flat.copy_(interaction)
            self.graph.replay()
            if hasattr(self.optimizer, 'on_replay'):
                self.optimizer.on_replay()
            return self.loss
        same = all(k in interaction and interaction[k].shape == v.shape for k, v in self.static.items())
        if not same:
            return self._eager(interaction)
        keys = list(self.static)
        torch._foreach_copy_([self.static[k] for k in keys], [interaction[k] for k in keys])      # one launch per dtype, not one per field
        self.graph.replay()
        if hasattr(self.optimizer, 'on_replay'):
            self.optimizer.on_replay()            # host-side bookkeeping of optimizers whose update count lives on the device
        return self.loss
