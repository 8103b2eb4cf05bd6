"""The caller side of the hot path: one optimisation step as run_plnerf.py:1235-1316 performs it,
plus the checkpoint wire format (run_plnerf.py:1324-1332, 454-471).

`TrainStep` reproduces the reference loop body's semantics exactly -- loss = mse(rgb) + mse(rgb0),
two Adam optimisers (fine network / coarse network), exponential LR decay applied to BOTH
optimisers from the FINE learning rate (the reference assigns `new_lrate` to the coarse optimiser
too, line 1315), `constant_init` warm-up (line 1284) -- and adds what the reference does not have:
ray shards per rank with one bucketed gradient all-reduce (dp.py).  Ray selection happens on the
device (only the N_rand selected pixels are turned into rays) instead of rebuilding the full
H x W ray grid and choosing on the host every step (lines 1259, 1275).
"""
import os

import numpy as np
import torch

from . import dp
from .render import render


class _MseFn(torch.autograd.Function):
    """mean((x - y) ** 2) with a two-launch backward (autograd's chain for the expression is seven)."""

    @staticmethod
    def forward(ctx, x, y):
        d = x - y
        ctx.save_for_backward(d)
        return (d * d).mean()

    @staticmethod
    def backward(ctx, g):
        (d,) = ctx.saved_tensors
        gx = d * (g * (2.0 / d.numel()))
        return (gx if ctx.needs_input_grad[0] else None), (-gx if ctx.needs_input_grad[1] else None)


def img2mse(x, y):
    """run_nerf_helpers.py:17."""
    if x.shape != y.shape:
        return torch.mean((x - y) ** 2)     # broadcasting operands: leave the bookkeeping to autograd
    return _MseFn.apply(x, y)


def mse2psnr(x):
    return -10. * torch.log(x) / torch.log(torch.tensor(10., device=x.device))


def select_rays(H, W, K, c2w, n_rand, generator=None, precrop=None):
    """Rays of `n_rand` distinct random pixels of one view, built on c2w's device.
    Same ray convention as get_rays (run_nerf_helpers.py:162-171).  `precrop` = (dH, dW) limits the
    draw to the central 2dH x 2dW window (run_plnerf.py:1261-1270).  Returns (batch_rays [2,n,3],
    pixel rows, pixel cols)."""
    dev = c2w.device
    if precrop is not None:
        dH, dW = precrop
        r0, c0, nr, nc = H // 2 - dH, W // 2 - dW, 2 * dH, 2 * dW
    else:
        r0, c0, nr, nc = 0, 0, H, W
    pick = torch.randperm(nr * nc, generator=generator, device=dev)[:n_rand]
    rows = r0 + pick // nc
    cols = c0 + pick % nc
    cam = torch.stack([(cols.float() - K[0][2]) / K[0][0], -(rows.float() - K[1][2]) / K[1][1],
                       -torch.ones(n_rand, device=dev)], -1)
    rays_d = torch.sum(cam[:, None, :] * c2w[:3, :3], -1)
    rays_o = c2w[:3, -1].expand(rays_d.shape)
    return torch.stack([rays_o, rays_d], 0), rows, cols


class TrainStep:
    """One training iteration of the reference loop on the HIP path.

    render_kwargs / optimizer / optimizer_coarse come from create_nerf(args).  `args` supplies
    lrate, lrate_decay, constant_init (iterations of forced constant-mode warm-up), chunk."""

    def __init__(self, args, render_kwargs_train, optimizer, optimizer_coarse, start=0, distributed=None):
        self.args = args
        self.kw = render_kwargs_train
        self.optimizer = optimizer
        self.optimizer_coarse = optimizer_coarse
        self.global_step = start
        self.nets = [n for n in (self.kw["network_fn"], self.kw.get("network_fine")) if n is not None]
        distributed = torch.distributed.is_initialized() if distributed is None else distributed
        self.bucket = None
        if distributed and torch.distributed.get_world_size() > 1:
            # replicas must start from the same weights (create_nerf initialises from each process's own RNG, and a
            # checkpoint may have been loaded on one rank only): averaged gradients applied to different weights
            # diverge silently
            dp.broadcast_parameters(self.nets)
            self.bucket = dp.GradientBucket(self.nets)

    def learning_rate(self):
        decay_rate, decay_steps = 0.1, self.args.lrate_decay * 1000
        return self.args.lrate * (decay_rate ** (self.global_step / decay_steps))

    def __call__(self, H, W, K, batch_rays, target_s, near=0., far=1.):
        i = self.global_step + 1                      # the reference iterates i = start+1 .. N_iters
        chunk = getattr(self.args, "chunk", 1024 * 32)
        rgb, disp, acc, extras = render(H, W, K, chunk=chunk, rays=batch_rays, near=near, far=far, retraw=True,
                                        constant_init=i < getattr(self.args, "constant_init", 0), **self.kw)
        self.optimizer.zero_grad()
        self.optimizer_coarse.zero_grad()
        img_loss = img2mse(rgb, target_s)
        loss = img_loss
        psnr = mse2psnr(img_loss.detach())
        if 'rgb0' in extras:
            loss = loss + img2mse(extras['rgb0'], target_s)
        loss.backward()
        if self.bucket is not None:
            self.bucket.allreduce_mean()
        self.optimizer.step()
        self.optimizer_coarse.step()
        new_lrate = self.learning_rate()
        for group in self.optimizer.param_groups:
            group['lr'] = new_lrate
        for group in self.optimizer_coarse.param_groups:
            group['lr'] = new_lrate                   # sic: the reference uses the fine rate here (line 1315)
        self.global_step += 1
        return loss.detach(), psnr


def save_checkpoint(path, global_step, network_fn, network_fine, optimizer):
    """The reference's checkpoint dict (run_plnerf.py:1324-1332): note that only the FINE network's
    optimizer state is stored there (the coarse optimizer restarts from scratch on resume)."""
    torch.save({
        'global_step': global_step,
        'network_fn_state_dict': network_fn.state_dict(),
        'network_fine_state_dict': network_fine.state_dict() if network_fine is not None else None,
        'optimizer_state_dict': optimizer.state_dict(),
    }, path)


def checkpoint_path(basedir, expname, step):
    """'{:06d}.tar' under ckpt_dir/expname, the name create_nerf's reload scans for."""
    return os.path.join(basedir, expname, '{:06d}.tar'.format(step))
