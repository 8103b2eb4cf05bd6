"""Ray-sharded data parallelism: one process per GPU, rays split evenly, one all-reduce of
the flattened MLP gradients per step (RCCL over xGMI with backend "nccl"; gloo on CPU).

The reference has no distributed path in run_plnerf.py (its depth variant wraps the model
in single-process nn.DataParallel, run_nerf_sample_based_depth.py:564,585); SURVEY.md
section 8e specifies this design instead: every op on the path is per ray, the only
cross-ray coupling is the mean in img2mse, so with equal shards the global gradient is the
average of the per-rank gradients.  Payload: 2 networks x 595,844 fp32 = 4.77 MB, sent as ONE
in-place all-reduce per network (2.38 MB each, latency-bound on the point-to-point xGMI links:
few large messages, not 48 small ones), the fine network's issued while the coarse network's
backward is still running.
"""
import os

import torch
import torch.distributed as dist


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
    the gradient, the autograd stream goes on).  The fine network's backward finishes first, so its 2.4 MB travel
    over xGMI while the coarse network's backward still computes; `allreduce_mean()` -- called where the reference
    loop would step the optimizers -- waits for both and applies the 1/world factor.

    Gradients in any other layout (another module, a CPU test) fall back to one gathered bucket per call."""

    def __init__(self, modules, group=None, overlap=True):
        self.modules = list(modules)
        self.group = group
        self.params = [p for m in self.modules for p in m.parameters()]
        self.numel = sum(p.numel() for p in self.params)
        self._owner = {}
        self._arrived = [0] * len(self.modules)
        self._works = []         # (work handle, flat tensor) of the collectives in flight
        self._reduced = [False] * len(self.modules)
        self._hooks = []
        self.flat = None         # fallback bucket, allocated on first use
        if overlap and dist.is_initialized() and dist.get_world_size(group) > 1:
            for mi, m in enumerate(self.modules):
                for p in m.parameters():
                    self._owner[p] = mi
                    self._hooks.append(p.register_post_accumulate_grad_hook(self._on_grad))

    # -- overlapped path ---------------------------------------------------------------------------------------
    def _on_grad(self, p):
        mi = self._owner[p]
        self._arrived[mi] += 1
        if self._arrived[mi] == sum(1 for _ in self.modules[mi].parameters()):
            self._arrived[mi] = 0
            self._launch(mi)

    def _launch(self, mi):
        from .optim import flat_view_of
        flat = flat_view_of([p.grad for p in self.modules[mi].parameters()])
        if flat is None:
            return                                        # not one buffer: the fallback handles this module
        work = dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        self._works.append((work, flat))
        self._reduced[mi] = True

    def pending(self):
        """Collectives enqueued by the hooks and not yet waited for (for tests / diagnostics)."""
        return len(self._works)

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

    def allreduce_mean(self, group=None, force=False):
        """Average every gradient over the ranks (in place).  Returns the number of collectives used."""
        group = group if group is not None else self.group
        world = dist.get_world_size(group) if dist.is_initialized() else 1
        if world == 1 and not (force and dist.is_initialized()):
            return 0
        if not self._hooks:                               # single rank with force=True, or overlap disabled
            for mi in range(len(self.modules)):
                self._launch(mi)
        n = len(self._works)
        # the EXPOSED part of the exchange, as the launch stream sees it: from here (the backward's last kernel is
        # enqueued) until both collectives have been waited for and scaled.  Events only while bench.py has a timer
        # installed; they are recorded on the current stream, which work.wait() makes wait for RCCL's stream.
        from . import functional as Fn
        ev = None
        if Fn.KERNEL_TIMER is not None and self._works and self._works[0][1].is_cuda:
            ev = Fn.KERNEL_TIMER.bracket("allreduce_exposed")
            ev[0].record()
        for work, flat in self._works:
            work.wait()
            flat.mul_(1.0 / world)
        if ev is not None:
            ev[1].record()
        self._works = []
        self._arrived = [0] * len(self.modules)           # (a module with parameters that never receive a gradient never completes)
        rest = [m for mi, m in enumerate(self.modules) if not self._reduced[mi]]
        self._reduced = [False] * len(self.modules)
        if rest:
            self._fallback(rest, world)
            n += 1
        return n


def broadcast_parameters(modules, src=0, group=None):
    """Make every rank start from rank `src`'s weights (replicas then stay bit-identical,
    since each applies the same averaged gradient with the same optimizer state)."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return
    for m in modules:
        for p in m.parameters():
            dist.broadcast(p.data, src=src, group=group)
