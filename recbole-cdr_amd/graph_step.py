"""hipGraph capture of one drop-in training step (calculate_loss -> backward -> dense Adam).

On small, L2-resident configurations (BASELINE C1-C4) the reference's step is hundreds of tiny launches; ours is fewer
but still launch/host bound from Python.  Every native call enqueues on torch's current stream and allocates through
torch's caching allocator, so the whole step can be captured once into a hipGraph and replayed: the host then pays one
graph launch per step instead of one ctypes call + one allocation per kernel.

Where the batch comes from:
  * ``producer`` (data/producer.py): the batch is produced by a launch INSIDE the captured step into the producer's own fixed
    tensors (device cursor, device-side negative sampling) -- ``replay()`` is the whole step, nothing is copied;
  * otherwise the batch tensors are static buffers that ``step`` copies the new ids / labels into (same shapes every step; a ragged
    last batch runs eagerly).
``loss_sum`` (a device scalar): every step -- replayed or eager -- adds its loss to it, inside the graph for replays, so an epoch's
loss total costs no extra launch and no host sync per step (``CrossDomainTrainer`` reads it once per epoch).
"""
import os
import torch

from .binding import capturing
from .data.interaction import Interaction
from .utils import total_loss as _total


def dev_of(interaction):
    return next(iter(interaction.values())).device


def step_and_sum(optimizer, loss, loss_sum):
    """``optimizer.step()`` + ``loss_sum += loss``.  trainer.DenseAdam does the addition inside its own counter launch (``loss_pair``);
    with any other optimizer -- or when no parameter had a gradient -- it is one elementwise launch here."""
    fold = loss_sum is not None and hasattr(optimizer, 'loss_pair') and loss.dtype == torch.float32 and loss.dim() == 0 and loss.is_contiguous()
    if fold:
        optimizer.loss_pair = (loss, loss_sum)
    optimizer.step()
    if loss_sum is not None and (not fold or optimizer.loss_pair is not None):
        if fold:
            optimizer.loss_pair = None
        loss_sum.add_(loss)


class GraphedTrainStep:
    def __init__(self, model, optimizer, example_interaction=None, warmup=3, restore_after_warmup=True, producer=None, loss_sum=None, unroll=1,
                 pipeline=True):
        """``restore_after_warmup``: the warm-up runs REAL steps on ``example_interaction`` (every lazily created buffer, native context
        and optimizer state must exist before the capture); with True the parameters and the optimizer state are put back afterwards,
        so swapping the eager loop for the graphed one does not add ``warmup`` extra updates on batch 0."""
        self.model, self.optimizer = model, optimizer
        self.restore_after_warmup = restore_after_warmup
        self.producer, self.loss_sum = producer, loss_sum
        if producer is not None:
            self.static = producer.fields
            self.flat, self._layout = None, None
            dev = next(iter(self.static.values())).device
        else:
            # the batch fields live side by side in ONE byte buffer (16-B aligned views): a caller that hands over a batch already
            # packed this way (``pack``) costs one device copy per step instead of one per field
            self._layout, off = {}, 0
            for k, v in example_interaction.items():
                nb = v.numel() * v.element_size()
                self._layout[k] = (off, nb, v.dtype, tuple(v.shape))
                off += (nb + 15) // 16 * 16
            dev = next(iter(example_interaction.values())).device
            self.flat = torch.empty(max(off, 16), device=dev, dtype=torch.uint8)
            self.static = Interaction()
            for k, (o, nb, dt, shape) in self._layout.items():
                self.static[k] = self.flat[o:o + nb].view(dt).view(shape)
            if getattr(example_interaction, 'k_major', None) is not None:
                self.static.k_major = example_interaction.k_major          # the layout hint is part of what the capture was made for
            for k, v in example_interaction.items():
                self.static[k].copy_(v)
        # ``unroll`` (with a producer only): a second graph holding ``unroll`` consecutive steps.  Between two graph launches the GPU idles
        # for ~5-9 us (measured: profiles/r04_graph_gap.txt) -- a fifth of a 40 us step at the reference's default batch; ``replay_many``
        # pays it once per ``unroll`` steps.
        self.unroll = int(unroll) if producer is not None else 1
        self.pipeline = pipeline             # True: pipelined one batch ahead | 'two_ahead' | False: the unrolled graph keeps the plain launch order
        self._statics = None
        self._side2 = torch.cuda.Stream(device=dev) if producer is not None else None
        self.graph = None
        self.graph_k = None
        self.loss = None
        self._one = torch.ones((), device=dev, dtype=torch.float32)       # d loss / d loss, made once: backward() would fill one per step
        self._capture(warmup)

    def _eager(self, interaction):
        # set_to_none: autograd then hands each parameter its gradient buffer instead of "zero-fill, then add into it" (two
        # extra kernels per parameter per step); inside the capture those buffers come from the graph's private pool, so
        # their addresses are the same on every replay
        self.optimizer.zero_grad(set_to_none=True)
        losses = self.model.calculate_loss(interaction)
        loss = _total(losses)
        if loss.dim():
            loss = loss.reshape(()) if loss.numel() == 1 else loss.sum()   # (a [1]-shaped loss: a view, not a reduction launch)
        loss.backward(self._one if loss.dtype == torch.float32 else None)
        loss = loss.detach()
        step_and_sum(self.optimizer, loss, self.loss_sum)
        return loss

    def _whole(self):
        """What the graph holds: (produce the batch) -> step -> (add the loss to the epoch's total)."""
        if self.producer is not None:
            self.producer.launch()
        return self._eager(self.static)

    def _can_pipeline(self):
        return (self.producer is not None and self.unroll > 1 and hasattr(self.model, 'prepare_batch')
                and hasattr(self.model, 'apply_rows_early') and getattr(self.optimizer, 'row_opt', None) is not None
                and bool(self.pipeline))

    def _whole_many(self, k):
        """``k`` consecutive steps.  With a model that offers ``prepare_batch`` / ``apply_rows_early`` (CoNet on the deferred Adam) the
        steps are SOFTWARE-PIPELINED over two streams: behind step i's forward launch the side stream runs {row update of step i ->
        produce batch i+1 -> its id sort -> the replay of its rows' postponed updates} while the main stream runs {weight gradients ->
        their reduction -> dense Adam of the tower weights}; they join before step i+1's forward.  The same launches on the same operands
        as the plain order (only independent launches overlap): bit-identical results.  Dependencies: the producer overwrites the batch
        buffers after the forward that read them (the backward works on the forward's own copies); the next batch's replay comes
        after this batch's row update on the same stream; the next forward needs both branches."""
        if not self._can_pipeline():
            # Round 6: with an optimizer that can host the production (trainer.DenseAdam.produce_jobs -> cdr_adam_multi_dev_produce) batch
            # i + 1 is produced INSIDE step i's optimizer launch -- the two are independent, 6-8 us each at the reference's batch -- so an
            # unrolled graph holds one stand-alone producer launch instead of k.  Same launches' work on the same operands: bit-identical.
            fuse = (self.producer is not None and k > 1 and hasattr(self.producer, 'jobs') and hasattr(self.optimizer, 'produce_jobs')
                    and getattr(self.optimizer, 'row_opt', None) is None and self.pipeline is not False)
            loss = None
            if fuse:
                self.producer.launch()
                for i in range(k):
                    if i + 1 < k:
                        self.optimizer.produce_jobs = self.producer.jobs()
                    loss = self._eager(self.static)
                return loss
            for _ in range(k):
                loss = self._whole()
            return loss
        main, side = torch.cuda.current_stream(), self._side2
        if self.pipeline != 'two_ahead' or not (hasattr(self.model, 'sort_batch') and hasattr(self.producer, 'add_slot')):
            return self._whole_many_one_ahead(k, main, side)
        # pipeline='two_ahead' (measured equal to the default on C3 -- the two branches do not overlap enough for the shorter critical
        # path to show, DESIGN 4.R4 -- and kept selectable).  The production and the id sort of a batch depend on nothing a step writes, so they are issued TWO steps
        # ahead at the tail of the main stream (behind the dense Adam, where the main branch has slack), and the side stream keeps only
        # what depends on the row update: {row update of step i -> replay of batch i+1's rows}.  Critical path of a step: forward +
        # max(weight gradients + reduction + dense Adam + producer + sort, row update + replay) instead of forward + row update +
        # producer + sort + replay.  Two batch slots (batch i+2 overwrites batch i, whose forward is behind it on the main stream) and
        # three sets of sort buffers (the row update of step i still reads set i while set i+2 is written).
        P, M = self.producer, self.model
        if self._statics is None:
            self._statics = [P.fields_slot(0), P.fields_slot(P.add_slot())]
        st = self._statics
        P.launch(0); M.sort_batch(st[0], 0); M.replay_batch(st[0], 0)
        if k > 1:
            P.launch(1); M.sort_batch(st[1], 1)
        loss = None
        for i in range(k):
            cur = st[i & 1]
            self.optimizer.zero_grad(set_to_none=True)
            losses = M.calculate_loss(cur)
            loss = _total(losses)
            if loss.dim():
                loss = loss.reshape(()) if loss.numel() == 1 else loss.sum()
            side.wait_stream(main)
            with torch.cuda.stream(side):
                ahead = bool(M.apply_rows_early())
                if ahead and i + 1 < k:
                    M.replay_batch(st[(i + 1) & 1], (i + 1) % 3)
            loss.backward(self._one)
            loss = loss.detach()
            step_and_sum(self.optimizer, loss, self.loss_sum)
            if i + 2 < k:
                P.launch(i & 1); M.sort_batch(cur, (i + 2) % 3)
            main.wait_stream(side)
            if not ahead and i + 1 < k:                      # the row update could not run early: the next batch's replay waits for it
                M.replay_batch(st[(i + 1) & 1], (i + 1) % 3)
        return loss

    def _whole_many_one_ahead(self, k, main, side):
        """The default pipelined order: the side stream runs {row update of step i, which also produces batch i+1 -> the replay of its rows
        in one launch with its id sort's counting pass -> the sort's scatter} beside the main stream's weight gradients (+ loss total) and
        dense Adam (round 6: three launches on that chain, six before)."""
        self.producer.launch()
        self.model.prepare_batch(self.static)
        ro = getattr(self.optimizer, 'row_opt', None)
        try:
            # every loss of this sequence is differentiated at once and read only by the optimizer's launch behind the backward: the model's
            # forward may leave the addition of its loss partials to its backward's launches (CoNet: cdr_conet_defer_finish, one launch less)
            if ro is not None and os.environ.get('CDR_CONET_DEFER_FINISH', '1') != '0':
                ro.defer_finish = True
            return self._one_ahead_steps(k, main, side)
        finally:
            if ro is not None:
                ro.defer_finish = False

    def _one_ahead_steps(self, k, main, side):
        loss = None
        for i in range(k):
            self.optimizer.zero_grad(set_to_none=True)
            losses = self.model.calculate_loss(self.static)
            loss = _total(losses)
            if loss.dim():
                loss = loss.reshape(()) if loss.numel() == 1 else loss.sum()
            side.wait_stream(main)
            ro = getattr(self.optimizer, 'row_opt', None)
            with torch.cuda.stream(side):
                # the row update's launch also produces batch i + 1 (lazyadam.DeferredRowAdam.produce_jobs: one launch less on this chain)
                fused = (ro is not None and i + 1 < k and hasattr(ro, 'produce_jobs') and hasattr(self.producer, 'jobs')
                         and os.environ.get('CDR_APPLY_PRODUCE', '1') != '0')
                if fused:
                    ro.produce_jobs = self.producer.jobs()
                ahead = bool(self.model.apply_rows_early())
                produced = fused and ro.produce_jobs is None
                if ro is not None and hasattr(ro, 'produce_jobs'):
                    ro.produce_jobs = None
                if ahead and i + 1 < k:
                    if not produced:
                        self.producer.launch()
                    self.model.prepare_batch(self.static)
            loss.backward(self._one)
            loss = loss.detach()
            step_and_sum(self.optimizer, loss, self.loss_sum)
            main.wait_stream(side)
            if not ahead and i + 1 < k:                      # the row update could not run early: the next batch waits for it
                self.producer.launch()
                self.model.prepare_batch(self.static)
        return loss

    def _capture(self, warmup):
        """Warm-up (real steps, state put back afterwards), then the captures.  Whatever fails on the way -- a warm-up step, the pipelined
        pre-run, a capture -- the parameters, optimizer state and loader / sampler counters are restored to what they were on entry
        before the exception leaves (a failed capture used to leave the warm-up's updates applied), and the exception
        carries ``state_restored`` so that the trainer's warning can say so."""
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        snap = self._snapshot() if self.restore_after_warmup else None
        try:
            # populate .grad and the optimizer state (and every lazily created native context) before capture
            with torch.cuda.stream(side):
                for i in range(max(warmup, 1)):
                    self._whole()
                    if i == 0 and snap is not None:
                        snap = self._snapshot(snap)           # optimizer state created by the first step: remembered as zeros
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            if snap is not None:
                self._restore(snap)
            self.graph = torch.cuda.CUDAGraph()
            # capture on the stream that ran the warm-up: the native contexts are per (device, stream) and creating one
            # allocates, which is not allowed while a stream is capturing
            with capturing(self.graph, side):
                self.loss = self._whole()
            if self.unroll > 1:
                if self._can_pipeline():
                    # the pipelined order once eagerly (on the capture stream): the side stream's native context must exist before a
                    # capture uses it, and the state it moves is put back like the warm-up's
                    with torch.cuda.stream(side):
                        with torch.cuda.stream(self._side2):
                            from . import binding as B_
                            B_.ctx(dev_of(self.static))
                        self._whole_many(2)
                    torch.cuda.synchronize()
                    if snap is not None:
                        self._restore(snap)
                self.graph_k = torch.cuda.CUDAGraph()
                with capturing(self.graph_k, side):
                    self._whole_many(self.unroll)
        except BaseException as e:
            restored = False
            try:
                torch.cuda.synchronize()
                if snap is not None:
                    self._restore(snap)
                    restored = True
            except Exception:                                 # noqa: BLE001 -- a stream left in an invalidated capture: nothing more to do here
                pass
            self.graph = self.graph_k = None
            e.state_restored = restored
            raise

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
        if hasattr(self.model, 'graph_state_tensors'):
            ts += list(self.model.graph_state_tensors())     # device-side counters a model's loss advances (e.g. an in-loss sampler's)
        if self.producer is not None:
            ts += self.producer.state_tensors()
        if self.loss_sum is not None:
            ts.append(self.loss_sum)
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

    def replay(self):
        """One captured step (with a producer: on the NEXT batch of the loader).  The caller keeps the loader's position in step
        (``producer.advance()``)."""
        self.graph.replay()
        if hasattr(self.optimizer, 'on_replay'):
            self.optimizer.on_replay()            # host-side bookkeeping of optimizers whose update count lives on the device
        return self.loss

    def replay_many(self):
        """``unroll`` captured steps in one graph launch (producer only: each of them produces its own batch)."""
        self.graph_k.replay()
        if hasattr(self.optimizer, 'on_replay'):
            for _ in range(self.unroll):
                self.optimizer.on_replay()

    def matches(self, interaction):
        return all(k in interaction and interaction[k].shape == v.shape and interaction[k].dtype == v.dtype for k, v in self.static.items()) \
            and getattr(interaction, 'k_major', None) == getattr(self.static, 'k_major', None)

    def step(self, interaction):
        if torch.is_tensor(interaction):                      # a batch packed by ``pack``
            assert interaction.dtype == torch.uint8 and interaction.numel() == self.flat.numel(), 'not a batch packed for this step'
            self.flat.copy_(interaction)
            return self.replay()
        if self.producer is not None or not self.matches(interaction):
            return self._eager(interaction)                   # a batch of another shape (ragged tail), or one the loader itself served
        keys = list(self.static)
        torch._foreach_copy_([self.static[k] for k in keys], [interaction[k] for k in keys])      # one launch per dtype, not one per field
        return self.replay()


class GraphedRowwiseStep:
    """The rowwise counterpart (optimizer_mode='rowwise': tables too large for dense gradients): {device batch producer ->
    ``model.fused_train_step`` -> loss total} x ``unroll`` captured as one hipGraph.  There is no warm-up to undo -- at 72 GB of tables
    nothing can be snapshotted -- so the caller runs its first real steps of the phase through ``eager()`` (on this object's stream: the
    native contexts and every lazily created buffer then exist where the capture needs them) and only then calls ``capture()``, which
    executes nothing.  The model keeps its host-side update counts in step through ``model.fused_replayed(n)``."""

    def __init__(self, model, producer, loss_sum, step_kwargs, unroll=4):
        self.model, self.producer, self.loss_sum, self.kw, self.unroll = model, producer, loss_sum, dict(step_kwargs), int(unroll)
        self.side = torch.cuda.Stream(device=dev_of(producer.fields))
        self.graph = self.graph_k = None
        self.eager_steps = 0

    def _one(self):
        self.producer.launch()
        loss = self.model.fused_train_step(self.producer.fields, **self.kw)
        self.loss_sum.add_(loss.detach().reshape(()))

    def eager(self):
        """One real step on the capture stream (ordered with the caller's stream on both sides)."""
        cur = torch.cuda.current_stream()
        self.side.wait_stream(cur)
        with torch.cuda.stream(self.side):
            self._one()
        cur.wait_stream(self.side)
        self.eager_steps += 1

    def capture(self):
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        with capturing(self.graph, self.side):
            self._one()
        if self.unroll > 1:
            self.graph_k = torch.cuda.CUDAGraph()
            with capturing(self.graph_k, self.side):
                for _ in range(self.unroll):
                    self._one()

    def replay(self, many=False):
        (self.graph_k if many else self.graph).replay()
        self.model.fused_replayed(self.unroll if many else 1)
