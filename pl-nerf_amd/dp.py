"""Ray-sharded data parallelism: one process per GPU, rays split evenly, one all-reduce of
the flattened MLP gradients per step (RCCL over xGMI with backend "nccl"; gloo on CPU).

The reference has no distributed path in run_plnerf.py (its depth variant wraps the model
in single-process nn.DataParallel, run_nerf_sample_based_depth.py:564,585); SURVEY.md
section 8e specifies this design instead: every op on the path is per ray, the only
cross-ray coupling is the mean in img2mse, so with equal shards the global gradient is the
average of the per-rank gradients.  Payload: 2 networks x 595,844 fp32 = 4.77 MB, sent as a
single bucket -- at this size a collective is latency-bound, so fewer, larger messages win
on the point-to-point xGMI links.
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
    """One flat fp32 buffer holding every parameter gradient of the given modules, in
    parameter order; `allreduce_mean()` averages it across ranks in a single collective and
    scatters the result back into the `.grad` tensors."""

    def __init__(self, modules):
        self.params = [p for m in modules for p in m.parameters()]
        self.numel = sum(p.numel() for p in self.params)
        dev = self.params[0].device
        self.flat = torch.zeros(self.numel, device=dev, dtype=torch.float32)
        self.views = []
        off = 0
        for p in self.params:
            self.views.append(self.flat[off:off + p.numel()].view_as(p))
            off += p.numel()

    def _pieces(self):
        """The gradients as few flat tensors as their layout allows: functional.MlpFn.backward returns a network's
        24 gradients as slices of one buffer, so this is normally one piece per network."""
        from .optim import contiguous_runs
        runs = contiguous_runs([p.grad for p in self.params])
        return None if any(g is None for _, _, g in runs) else [g for _, _, g in runs]

    def gather(self):
        """All .grad tensors -> the flat bucket, in ONE kernel (torch.cat into the buffer)."""
        if all(p.grad is not None for p in self.params):
            self._scatter_to = self._pieces()
            torch.cat(self._scatter_to if self._scatter_to is not None else
                      [p.grad.reshape(-1) for p in self.params], out=self.flat)
            return
        self._scatter_to = None
        for p, v in zip(self.params, self.views):
            if p.grad is None:
                v.zero_()
            else:
                v.copy_(p.grad)

    def scatter(self):
        """Flat bucket -> the .grad tensors, in one multi-tensor kernel."""
        pieces = getattr(self, "_scatter_to", None)
        if pieces is not None:
            torch._foreach_copy_(pieces, list(self.flat.split([t.numel() for t in pieces])))
            return
        missing = [p for p in self.params if p.grad is None]
        for p in missing:
            p.grad = torch.empty_like(p)
        torch._foreach_copy_([p.grad for p in self.params], self.views)

    def allreduce_mean(self, group=None, force=False):
        world = dist.get_world_size(group) if dist.is_initialized() else 1
        if world == 1 and not (force and dist.is_initialized()):
            return
        self.gather()
        dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=group)
        self.flat.mul_(1.0 / world)
        self.scatter()


def broadcast_parameters(modules, src=0, group=None):
    """Make every rank start from rank `src`'s weights (replicas then stay bit-identical,
    since each applies the same averaged gradient with the same optimizer state)."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return
    for m in modules:
        for p in m.parameters():
            dist.broadcast(p.data, src=src, group=group)
