"""Guarded bring-up of a multi-GPU layout: make the first hardware run un-losable.

``bench.py --gpus N`` (and anything else that builds RCCL sub-groups, several communicators on several streams, collectives on
views) has only ever run its N > 1 path over gloo on CPU and over RCCL with ONE rank.  The first real run may die -- or hang -- in
group creation or in the first collective of a layout.  ``try_layouts`` walks a chain of candidate layouts; each candidate is
built and driven through its first step inside a watchdog thread with a deadline, every rank's verdict is agreed over a gloo
CONTROL group (which never touches RCCL), and the first candidate that comes up on every rank is used.  When none does, the caller
falls back to N independent replicas (no data-path communication at all).  The record of what failed, with every rank's error
string, goes into the benchmark's JSON line (``layout_fallback``).

Fault injection for the CPU tests: ``CDR_PREFLIGHT_FAIL`` = comma-separated ``name[:raise|hang][@rank]`` entries.
"""
import os
import threading
import time

import torch
import torch.distributed as dist


def control_group(timeout_s=120):
    """A gloo group over all ranks, created through the store (no RCCL involved): barriers, verdicts and timings of the guarded phases."""
    import datetime
    return dist.new_group(backend='gloo', timeout=datetime.timedelta(seconds=timeout_s))


def _injected(name, rank):
    for ent in filter(None, os.environ.get('CDR_PREFLIGHT_FAIL', '').split(',')):
        spec, _, at = ent.partition('@')
        nm, _, how = spec.partition(':')
        if nm == name and (not at or int(at) == rank):
            return how or 'raise'
    return None


def run_guarded(fn, seconds, device=None):
    """``fn()`` in a daemon thread with a deadline -> (value, None) or (None, error string).  A blocked call (a communicator that
    never forms) cannot be cancelled; the thread is left behind and the caller must not touch what it was building."""
    box = {}

    def target():
        try:
            if device is not None and torch.device(device).type == 'cuda':
                torch.cuda.set_device(device)                       # (the current device is per thread)
            box['value'] = fn()
        except BaseException as e:                                  # noqa: BLE001 -- reported to the caller
            box['error'] = '%s: %s' % (type(e).__name__, str(e)[:400])
    th = threading.Thread(target=target, daemon=True)
    th.start()
    th.join(seconds)
    if th.is_alive():
        return None, 'no answer after %.0f s (still blocked)' % seconds
    if 'error' in box:
        return None, box['error']
    return box.get('value'), None


def agree(ok, err, ctrl):
    """Every rank's verdict -> (all ok?, {rank: error string} of the ranks that failed); over the gloo control group."""
    world = dist.get_world_size(ctrl)
    flag = torch.tensor([1 if ok else 0], dtype=torch.int32)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=ctrl)
    errs = [None] * world
    dist.all_gather_object(errs, err, group=ctrl)
    return bool(flag.item()), {r: e for r, e in enumerate(errs) if e}


def try_layouts(candidates, build, first_step, ctrl, seconds=30.0, device=None, cleanup=None):
    """``candidates``: names in order of preference.  ``build(name)`` -> layout object (creates its process groups and step objects);
    ``first_step(layout)`` runs one step and synchronises the device.  Returns (name, layout, attempts) with name = None when every
    candidate failed somewhere; ``attempts`` = [{'layout', 'ok', 'seconds', 'errors': {rank: str}}]."""
    rank = dist.get_rank(ctrl)
    attempts = []
    for name in candidates:
        t0 = time.perf_counter()

        def bring_up(name=name):
            how = _injected(name, rank)
            if how == 'raise':
                raise RuntimeError('injected failure of layout %r on rank %d' % (name, rank))
            if how == 'hang':
                time.sleep(3600)
            lay = build(name)
            first_step(lay)
            return lay
        lay, err = run_guarded(bring_up, seconds, device)
        ok, errs = agree(err is None, err, ctrl)
        attempts.append({'layout': name, 'ok': ok, 'seconds': time.perf_counter() - t0, 'errors': errs})
        if ok:
            return name, lay, attempts
        if cleanup is not None and err is None:
            cleanup(lay)                                            # came up here but not everywhere: give the memory back
        del lay
        if device is not None and torch.device(device).type == 'cuda':
            import gc
            gc.collect()
            torch.cuda.empty_cache()
    return None, None, attempts


def ranks_seen(group, device):
    """The size of ``group`` as the collective library itself counts it: an all-reduce of ones."""
    one = torch.ones(1, device=device, dtype=torch.float32)
    dist.all_reduce(one, group=group)
    return int(round(float(one.item())))
