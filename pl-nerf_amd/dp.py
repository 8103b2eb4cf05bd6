"""Ray-sharded data parallelism: one process per GPU, rays split evenly, one all-reduce of
the flattened MLP gradients per step (RCCL over xGMI with backend "nccl"; gloo on CPU).

The reference has no distributed path in run_plnerf.py (its depth variant wraps the model
in single-process nn.DataParallel, run_nerf_sample_based_depth.py:564,585); SURVEY.md
section 8e specifies this design instead: every op on the path is per ray, the only
cross-ray coupling is the mean in img2mse, so with equal shards the global gradient is the
average of the per-rank gradients.  Payload: 2 networks x 595,844 fp32 = 4.77 MB: ONE in-place
all-reduce over both networks' gradient buffers after the merged backward (they lie back to back;
latency-bound on the point-to-point xGMI links), or -- on the autograd-order route -- one per network
(2.38 MB each), the fine network's issued while the coarse network's backward is still running.
"""
import os

import torch
import torch.distributed as dist

from ._lib import GUARDED_PRECISIONS


def init_from_env(backend=None, force=False):
    """Initialise torch.distributed from RANK/WORLD_SIZE/MASTER_* (torchrun's contract).
    Returns (rank, world_size, local_rank).  No-op for a single process unless `force`
    (used to exercise the RCCL path on one GPU)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if (world > 1 or force) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def shard_rays(n_global, rank, world_size):
    """[begin, end) of this rank's contiguous, equal share of a global ray batch.  The mean
    over equal shards equals the global mean, so n_global must divide evenly."""
    if n_global % world_size != 0:
        raise ValueError(f"global batch {n_global} is not divisible by world size {world_size}")
    per = n_global // world_size
    return rank * per, (rank + 1) * per


class GradientBucket:
    """Gradient averaging across ranks, one collective per network, overlapped with the rest of the backward.

    functional.MlpFn.backward hands autograd a network's 24 gradients as consecutive slices of ONE buffer, and
    autograd keeps those views as the `.grad` tensors.  So a network's whole gradient is a single flat tensor that can
    be all-reduced IN PLACE -- no gather into a bucket, no scatter back.  A post-accumulate hook on every parameter
    counts arrivals; when a network is complete its all-reduce is enqueued at once (async: RCCL's stream waits for
    the stream the backward runs on, which goes on).  `finish(modules)` -- called where the reference loop would step
    that network's optimizer -- makes the current stream wait for those collectives.  What is left exposed after a
    network's last backward kernel is that ONE wait:

      * the 1 / world factor is not applied here when the caller can take it (`defer_scale`: optim.FlatAdam multiplies
        inside its step kernel) -- `.grad` then holds the SUM over ranks until the optimizer has stepped;
      * the networks' range status words (section 5 of DESIGN.md: a rank whose forward clamped has contributed a wrong
        gradient to the sum, so EVERY rank must withhold the step) travel as a tail element of the same buffer
        (functional.MlpFn.backward appends it; the weight-gradient reduction kernel writes it): after the SUM a non-zero
        tail means "some rank's forward left the half range" on every rank alike.  `tails()` hands those words to the
        optimizer as its guards -- no separate collective, no write-back launch.

    train.TrainStep's merged backward (both networks' backward as one launch sequence) assigns the gradients itself and
    calls `gradients_ready`: the two flat buffers lie back to back and go out as ONE collective.  Gradients in any other
    layout (another module, a CPU test) fall back to one gathered bucket per call."""

    TAIL = 4      # floats appended to a network's flat gradient by functional.MlpFn.backward ([0] = range status)

    def __init__(self, modules, group=None, overlap=True):
        self.modules = list(modules)
        self.group = group
        self.params = [p for m in self.modules for p in m.parameters()]
        self.numel = sum(p.numel() for p in self.params)
        self._owner = {}
        self._n_params = [sum(1 for _ in m.parameters()) for m in self.modules]
        self._arrived = [0] * len(self.modules)
        self._works = {}         # module index -> (work handle, reduced tensor, tail view or None)
        self._tails = {}         # module index -> tail view of the last finished exchange
        self._hooks = []
        self.flat = None         # fallback bucket, allocated on first use
        for m in self.modules:      # functional._grad_buffer then leaves each backward's flat buffer (with its tail) on the module
            m.__dict__["_wants_grad_flat"] = True
        if overlap and dist.is_initialized() and dist.get_world_size(group) > 1:
            for mi, m in enumerate(self.modules):
                for p in m.parameters():
                    self._owner[p] = mi
                    self._hooks.append(p.register_post_accumulate_grad_hook(self._on_grad))

    # -- overlapped path ---------------------------------------------------------------------------------------
    def _on_grad(self, p):
        mi = self._owner[p]
        self._arrived[mi] += 1
        if self._arrived[mi] == self._n_params[mi]:
            self._arrived[mi] = 0
            self._launch(mi)

    def _launch(self, mi):
        from .optim import flat_view_of
        m = self.modules[mi]
        flat = flat_view_of([p.grad for p in m.parameters()])
        if flat is None:
            return                                        # not one buffer: the fallback handles this module
        tail = None
        full = getattr(m, "_grad_flat", None)             # (functional.MlpFn.backward: the buffer with its status tail)
        if full is not None and full.data_ptr() == flat.data_ptr() and full.numel() == flat.numel() + self.TAIL:
            flat, tail = full, full[-self.TAIL:-self.TAIL + 1]
        work = dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        self._works[mi] = (work, flat, tail)

    def gradients_ready(self, modules):
        """A caller that assigned `.grad` itself (train.TrainStep's merged backward: no AccumulateGrad node ran, so no hook
        fired) says the gradients of `modules` are complete: their collectives are enqueued now, as the hooks would have."""
        if not self._hooks:
            return
        idx = [self._index(m) for m in modules]
        for mi in idx:
            self._arrived[mi] = 0
        idx = [mi for mi in idx if mi not in self._works]
        # the merged backward lays the networks' flat buffers (with their tails) back to back in one allocation: ONE
        # collective over the lot -- at 2.4 MB per network an all-reduce is latency, not bandwidth
        fulls = [getattr(self.modules[mi], "_grad_flat", None) for mi in idx]
        if len(idx) > 1 and all(f is not None for f in fulls):
            order = sorted(range(len(idx)), key=lambda k: fulls[k].data_ptr())
            first = fulls[order[0]]
            adjacent = all(fulls[order[k + 1]].data_ptr() == fulls[order[k]].data_ptr() + 4 * fulls[order[k]].numel() and
                           fulls[order[k]].untyped_storage().data_ptr() == first.untyped_storage().data_ptr()
                           for k in range(len(order) - 1))
            flats_ok = all(self._flat_with_tail(self.modules[mi]) is not None for mi in idx)
            if adjacent and flats_ok:
                total = sum(f.numel() for f in fulls)
                both = first.as_strided((total,), (1,))
                work = dist.all_reduce(both, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
                for mi, f in zip(idx, fulls):
                    self._works[mi] = (work, f, f[-self.TAIL:-self.TAIL + 1])
                return
        for mi in idx:
            self._launch(mi)

    def _flat_with_tail(self, m):
        """The module's gradient as ONE buffer with its status tail (functional._grad_buffer's layout), or None."""
        from .optim import flat_view_of
        flat = flat_view_of([p.grad for p in m.parameters()])
        full = getattr(m, "_grad_flat", None)
        if flat is None or full is None or full.data_ptr() != flat.data_ptr() or full.numel() != flat.numel() + self.TAIL:
            return None
        return full

    def pending(self):
        """Collectives enqueued by the hooks and not yet waited for (for tests / diagnostics)."""
        return len(self._works)

    def tails(self, modules=None):
        """The reduced range-status words (1-element float tensors, non-zero = some rank's forward of that network left
        the half range) of the exchanges `finish` completed last, for the given modules (default: all)."""
        idx = range(len(self.modules)) if modules is None else [self._index(m) for m in modules]
        return [self._tails[mi] for mi in idx if self._tails.get(mi) is not None]

    def _index(self, module):
        for mi, m in enumerate(self.modules):
            if m is module:
                return mi
        raise ValueError("GradientBucket: not one of this bucket's modules")

    # -- fallback: one gathered bucket ---------------------------------------------------------------------------
    def _fallback(self, modules, world):
        params = [p for m in modules for p in m.parameters()]
        if not params:
            return
        n = sum(p.numel() for p in params)
        if self.flat is None or self.flat.numel() < n or self.flat.device != params[0].device:
            self.flat = torch.zeros(self.numel, device=params[0].device, dtype=torch.float32)
        flat = self.flat[:n]
        views, off = [], 0
        for p in params:
            views.append(flat[off:off + p.numel()].view_as(p))
            off += p.numel()
        for p, v in zip(params, views):
            if p.grad is None:
                v.zero_()
            else:
                v.copy_(p.grad)
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group)
        flat.mul_(1.0 / world)
        for p, v in zip(params, views):
            if p.grad is None:
                p.grad = torch.empty_like(p)
            p.grad.copy_(v)

    def finish(self, modules=None, defer_scale=False, group=None, force=False):
        """Complete the exchange of `modules` (default: all of the bucket's): the current stream waits for their
        collectives.  Returns the factor the caller still owes the gradients: 1 / world if `defer_scale` and every one
        of them was reduced in place (`.grad` then holds the SUM), else 1.0 (scaled here: `.grad` holds the mean)."""
        group = group if group is not None else self.group
        world = dist.get_world_size(group) if dist.is_initialized() else 1
        idx = list(range(len(self.modules))) if modules is None else [self._index(m) for m in modules]
        self.collectives = 0
        if world == 1 and not (force and dist.is_initialized()):
            for mi in idx:      # (nothing to exchange: let go of the backward's flat buffers, as the exchange below would)
                self.modules[mi].__dict__.pop("_grad_flat", None)
            return 1.0
        for mi in idx:
            if mi not in self._works and not self._hooks:     # single rank with force=True, or overlap disabled
                self._launch(mi)
        mine = [mi for mi in idx if mi in self._works]
        rest = [mi for mi in idx if mi not in self._works]
        deferred = bool(defer_scale and not rest)
        # the EXPOSED part of the exchange, as the stream sees it: from here (the backward's last kernel is enqueued)
        # until the collectives have been waited for.  Events only while bench.py has a timer installed; they are
        # recorded on the current stream, which work.wait() makes wait for RCCL's stream.
        from . import functional as Fn
        ev = None
        if Fn.KERNEL_TIMER is not None and mine and self._works[mine[0]][1].is_cuda:
            last = (len(self.modules) - 1) in idx
            ev = Fn.KERNEL_TIMER.bracket("allreduce_exposed" if last else "allreduce_exposed_coarse")
            ev[0].record()
        waited = set()
        for mi in mine:
            work, flat, tail = self._works.pop(mi)
            work.wait()
            self.collectives -= id(work) in waited      # (one collective may carry several networks: gradients_ready)
            waited.add(id(work))
            self._tails[mi] = tail
            self.modules[mi].__dict__.pop("_grad_flat", None)      # (the tail view keeps the buffer alive as long as it is the guard)
            if not deferred:
                (flat if tail is None else flat[:-self.TAIL]).mul_(1.0 / world)
            self.collectives += 1
        if ev is not None:
            ev[1].record()
        for mi in idx:
            self._arrived[mi] = 0           # (a module with parameters that never receive a gradient never completes)
        if rest:
            for mi in rest:
                self._tails[mi] = None
            self._fallback([self.modules[mi] for mi in rest], world)
            self.collectives += 1
            self.collectives += self._sync_status(rest, group)
        return 1.0 / world if deferred else 1.0

    def _sync_status(self, idx, group):
        """The fallback exchange has no status tail (a batch split over several MlpFn calls, an optimizer that is not
        FlatAdam, overlap disabled: `.grad` is not the one buffer whose tail the reduction kernel writes).  The range guard
        is per network AND per rank, the averaged gradient is not: a rank whose forward clamped has contributed a wrong
        gradient, so EVERY rank must withhold the step, and every rank must raise at the next check_range() -- not one
        rank while the others block in the next collective.  One MAX all-reduce of the ranks' sticky status words (bit
        masks: MAX keeps "non-zero" and exists on every backend), written back into each network's own word and handed out by
        `tails()` as the optimizer's guards, exactly as on the overlapped path.  Returns the number of collectives used."""
        owners = [mi for mi in idx if _guarded(self.modules[mi])]
        if not owners:
            return 0
        words = [self.modules[mi].status_word() for mi in owners]
        both = torch.cat(words)
        dist.all_reduce(both, op=dist.ReduceOp.MAX, group=group)
        for mi, w, v in zip(owners, words, both.split(1)):
            w.copy_(v)      # (sticky on every rank: an optimizer guarded by the network itself, and check_range(), see it)
            self._tails[mi] = v
        return 1

    def allreduce_mean(self, group=None, force=False):
        """Average every gradient over the ranks (in place; `.grad` holds the mean afterwards).  Returns the number of
        collectives used."""
        self.finish(None, defer_scale=False, group=group, force=force)
        return self.collectives


def _guarded(m):
    """A network whose 16-bit kernels keep a range status word on the device (nerf.NeRF in a guarded precision)."""
    try:
        return (hasattr(m, "status_word") and getattr(m, "precision", None) in GUARDED_PRECISIONS and m.is_supported()
                and next(m.parameters()).is_cuda)
    except StopIteration:
        return False


def broadcast_optimizer_state(optimizers, src=0, group=None):
    """Rank `src`'s Adam moments and step counts to every rank (a checkpoint may have been restored on one rank only:
    replicas that share weights but not moments diverge at the first step).  optim.FlatAdam keeps a group's moments as
    two flat buffers: two broadcasts and one step count per group instead of three per parameter."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return
    for opt in optimizers:
        flats = getattr(opt, "_flat", None)
        for gi, g in enumerate(opt.param_groups):
            fl = flats[gi] if flats is not None and gi < len(flats) else None
            if fl is not None and all(opt.state.get(p) for p in g['params']):
                dist.broadcast(fl['m'], src=src, group=group)
                dist.broadcast(fl['v'], src=src, group=group)
                steps = torch.tensor([float(opt.state[p]['step']) for p in g['params']], dtype=torch.float32,
                                     device=fl['m'].device)
                dist.broadcast(steps, src=src, group=group)
                for p, v in zip(g['params'], steps.tolist()):
                    opt.state[p]['step'] = torch.tensor(float(v))
                fl.pop('uniform_step', None)
                continue
            for p in g['params']:
                st = opt.state.get(p)
                have = torch.tensor([1 if st else 0], device=p.device)
                dist.broadcast(have, src=src, group=group)
                if not int(have.item()):
                    continue
                if not st:      # this rank has no state yet: torch.optim.Adam's layout
                    st = opt.state[p] = {'step': torch.tensor(0.0), 'exp_avg': torch.zeros_like(p),
                                         'exp_avg_sq': torch.zeros_like(p)}
                for k in ('exp_avg', 'exp_avg_sq'):
                    dist.broadcast(st[k], src=src, group=group)
                step = torch.as_tensor(float(st['step']), device=p.device).reshape(1)
                dist.broadcast(step, src=src, group=group)
                st['step'] = torch.tensor(float(step.item()))


def broadcast_scalar(value, src=0, group=None, device=None):
    """Rank `src`'s Python number on every rank (the loop's global_step, hence its learning rate)."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return value
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.broadcast(t, src=src, group=group)
    return type(value)(t.item())


def broadcast_parameters(modules, src=0, group=None):
    """Make every rank start from rank `src`'s weights (replicas then stay bit-identical,
    since each applies the same averaged gradient with the same optimizer state).  A module whose parameters are the
    consecutive slices of one buffer (optim.FlatAdam re-homes them so) goes in ONE broadcast."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return
    from .optim import flat_view_of
    for m in modules:
        ps = [p.data for p in m.parameters()]
        flat = flat_view_of(ps) if ps and all(t.dtype == torch.float32 for t in ps) else None
        if flat is not None:
            dist.broadcast(flat, src=src, group=group)
            continue
        for p in ps:
            dist.broadcast(p, src=src, group=group)
