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
import ctypes
import os

import numpy as np
import torch

from . import _lib as L
from . import dp
from . import functional as Fn
from . import raybatch as RB
from .optim import FlatAdam
from .rays import ndc_rays
from .render import render, render_rays


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
    """run_nerf_helpers.py:18: -10 log10(x)."""
    return -10. * torch.log10(x)


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


def select_view_rays(H, W, K, c2w, image, n_rand, near, far, seed=0, step=0, ray_id0=0, precrop=None,
                     want_viewdirs=True, want_pixels=False):
    """Device-side version of the reference's per-step ray selection (run_plnerf.py:1259-1281) in ONE launch
    (plnerf_select_rays): `n_rand` distinct random pixels of the view `c2w` (inside the central precrop window, if
    given as (dH, dW)), their rays, unit view directions, near / far columns and target colours image[row, col].
    The choice is a keyed bijection of the pixel window evaluated at the global ray ids ray_id0 .. ray_id0 + n_rand
    - 1, so ranks with disjoint id ranges draw disjoint pixels of one global sample.
    Returns (RayColumns, target [n_rand, 3] or None, pixels [n_rand, 2] int32 or None)."""
    dev = image.device if image is not None else torch.device("cuda", torch.cuda.current_device())
    if precrop is not None:
        dH, dW = precrop
        r0, c0, nr, nc = H // 2 - dH, W // 2 - dW, 2 * dH, 2 * dW
    else:
        r0, c0, nr, nc = 0, 0, H, W
    c2w_host = (ctypes.c_float * 12)(*[float(v) for v in torch.as_tensor(c2w, device="cpu")[:3, :4].reshape(-1)])
    o, d = torch.empty(n_rand, 3, device=dev), torch.empty(n_rand, 3, device=dev)
    vd = torch.empty(n_rand, 3, device=dev) if want_viewdirs else None
    nr_col, fr_col = torch.empty(n_rand, device=dev), torch.empty(n_rand, device=dev)
    target = torch.empty(n_rand, 3, device=dev) if image is not None else None
    pix = torch.empty(n_rand, 2, device=dev, dtype=torch.int32) if want_pixels else None
    img_c = None if image is None else image.detach().to(torch.float32).contiguous()
    L.check(L.lib().plnerf_select_rays(
        int(H), int(W), float(K[0][0]), float(K[1][1]), float(K[0][2]), float(K[1][2]), c2w_host,
        L.dptr(img_c, "image"), r0, c0, nr, nc, int(seed), int(step), int(ray_id0), int(n_rand), float(near),
        float(far), L.dptr(o), L.dptr(d), L.dptr(vd), L.dptr(nr_col), L.dptr(fr_col), L.dptr(target),
        L.dptr(pix, "pixels", torch.int32), L.stream()), "plnerf_select_rays")
    return RB.RayColumns(o, d, nr_col, fr_col, vd), target, pix


def merged_backward_ok(tape, nets):
    """Can this step's two network backwards run as one launch sequence (backward_merged)?  One MlpFn forward per
    network (the batch fitted one launch each), coarse first, both networks native (NeRF.is_native), in one 16-bit
    precision and with one density activation, every parameter trainable, no gradient wanted for the networks' inputs."""
    if len(tape.outs) != 2 or len(nets) != 2 or nets[0] is nets[1]:
        return False
    ctxs = [o.grad_fn for o in tape.outs]
    if any(c is None or getattr(c, "saved_acts", None) is None for c in ctxs):
        return False
    if ctxs[0].net is not nets[0] or ctxs[1].net is not nets[1]:
        return False
    return all(n.is_native() and n.precision in L.GUARDED_PRECISIONS and all(p.requires_grad for p in n.parameters())
               for n in nets) and nets[0].precision == nets[1].precision and ctxs[0].beta == ctxs[1].beta and \
        not any(c.in_grad or c.n_cam for c in ctxs)


def backward_merged(tape, nets, roots, root_grads, bucket=None):
    """loss.backward() with the two networks' backward in one launch sequence: autograd from the loss's roots (the
    rendered images, the depth hypotheses) down to d loss / d raw of either network (plnerf_quad_bwd x 2, each leaving
    max |g_raw| as a by-product; autograd does not enter the MlpFn nodes), then functional.mlp_backward_multi; the
    gradients are assigned as autograd would have accumulated them into the zeroed `.grad`s, and a data-parallel bucket
    is told that the two networks' gradients are complete."""
    raw_c, raw_f = tape.outs
    Fn.ABSMAX_LOG = log = []
    try:
        g_raw_f, g_raw_c = torch.autograd.grad(tuple(roots), (raw_f, raw_c), tuple(root_grads))
    finally:
        Fn.ABSMAX_LOG = None
    grads_c, grads_f = Fn.mlp_backward_multi([raw_c, raw_f], [g_raw_c, g_raw_f], log)
    for net, grads in ((nets[0], grads_c), (nets[1], grads_f)):
        for p, g in zip(net.param_list(), grads):
            p.grad = g
    if bucket is not None:
        bucket.gradients_ready(nets)


class TrainStep:
    """One training iteration of the reference loop on the HIP path.

    render_kwargs / optimizer / optimizer_coarse come from create_nerf(args).  `args` supplies
    lrate, lrate_decay, constant_init (iterations of forced constant-mode warm-up), chunk.

    `.grad` after a step: the gradients autograd would have left -- except under data parallelism with optim.FlatAdam, where
    each network's flat gradient buffer is all-reduced in place and the 1 / world factor is applied INSIDE the step kernel:
    `.grad` then holds the SUM over the ranks (world times the mean).  Code that reads `.grad` after the step (gradient-norm
    logging, external clipping) divides by the world size, or calls `bucket.allreduce_mean()` itself before stepping.

    Random draws (stratified jitter, sampler u, pixel choice) come from a counter-based generator keyed on
    (`seed`, step, global ray id) (functional.DrawSource): a global batch gives the same step whether one rank
    renders it or N ranks render a shard each.  `counter_rng=False` restores torch.rand."""

    def __init__(self, args, render_kwargs_train, optimizer, optimizer_coarse, start=0, distributed=None, seed=0,
                 counter_rng=True, range_check_every=100, pipeline=None):
        """The step runs on ONE stream in the reference's order: render, loss, backward, both optimizers.  (Rounds 3-4
        also carried two-stream schedules -- the coarse network's chain beside the fine pass, and across step boundaries;
        bit-identical, measured +1.4 % / -0.8 % with twice the step jitter, profiles/r04_pipeline_ab.txt -- removed in
        round 5: the forward kernels and the weight-gradient kernel each fill a CU on their own, so two kernels "overlap"
        only by taking CUs from each other.)"""
        if pipeline is not None:      # (rounds 3-4's schedule argument: a caller written against them keeps working)
            import warnings
            warnings.warn("TrainStep(pipeline=...) is ignored: the two-stream schedules were removed in round 5 (one stream, "
                          "the reference's order)", DeprecationWarning, stacklevel=2)
        self.args = args
        self.kw = render_kwargs_train
        self.optimizer = optimizer
        self.optimizer_coarse = optimizer_coarse
        self.global_step = start
        self.nets = [n for n in (self.kw["network_fn"], self.kw.get("network_fine")) if n is not None]
        distributed = torch.distributed.is_initialized() if distributed is None else distributed
        self.rank, self.world = (torch.distributed.get_rank(), torch.distributed.get_world_size()) if distributed \
            else (0, 1)
        self.draws = Fn.DrawSource(seed=seed) if counter_rng else None
        # f16x3 / f16 clamp at the IEEE-half maximum.  The kernels record that in a status word and the guarded Adam
        # withholds the affected steps; the host looks every `range_check_every` steps (one 4-byte read per network)
        # and raises.  0 = never look (the caller does).
        self.range_check_every = int(range_check_every)
        self.seed = seed
        # Both networks' backward as ONE launch sequence (functional.mlp_backward_multi -> plnerf_mlp_bwd_multi): autograd
        # runs only as far as the two upstream gradients d loss / d raw (two plnerf_quad_bwd launches), then one gradient-
        # chain grid and one launch of each weight-gradient kernel cover both networks (round 5: -0.15 ms of a 6.4 ms step,
        # profiles/r05_merged_bwd_ab.txt).  Taken whenever the step has the reference's two native networks in a 16-bit
        # mode (_merged_backward_ok); anything else goes through torch.autograd.backward as before.  PLNERF_MERGED_BWD=0 in
        # the environment switches it off (A/B measurements).
        self.merged_backward = os.environ.get("PLNERF_MERGED_BWD", "1") != "0"
        self.merged_steps = 0      # steps that took the merged backward (the rest went through torch.autograd.backward)
        self.bucket = None
        if distributed and self.world > 1:
            # replicas must start from the same weights (create_nerf initialises from each process's own RNG, and a
            # checkpoint may have been loaded on one rank only): averaged gradients applied to different weights
            # diverge silently
            dp.broadcast_parameters(self.nets)
            dp.broadcast_optimizer_state([self.optimizer, self.optimizer_coarse])      # (moments, step counts)
            self.global_step = dp.broadcast_scalar(self.global_step, device=next(self.nets[0].parameters()).device)
            self.bucket = dp.GradientBucket(self.nets)

    def drain(self):
        """No-op (rounds 3-4: waited for the side streams' work in flight; everything is on the caller's stream now)."""

    def learning_rate(self):
        decay_rate, decay_steps = 0.1, self.args.lrate_decay * 1000
        return self.args.lrate * (decay_rate ** (self.global_step / decay_steps))

    def step_view(self, H, W, K, c2w, image, near=0., far=1., n_rand=None, precrop=None):
        """The loop body from the view on (run_plnerf.py:1259-1316): choose this rank's n_rand pixels of the view on
        the device, then the optimisation step.  `image` [H, W, 3] lives on the device."""
        n_rand = int(n_rand if n_rand is not None else self.args.N_rand)
        cols, target, _ = select_view_rays(H, W, K, c2w, image, n_rand, near, far, seed=self.seed,
                                           step=self.global_step, ray_id0=self.rank * n_rand, precrop=precrop,
                                           want_viewdirs=bool(self.kw.get("use_viewdirs", True)))
        return self._step(H, W, K, cols, target, near, far)

    def __call__(self, H, W, K, batch_rays, target_s, near=0., far=1.):
        return self._step(H, W, K, batch_rays, target_s, near, far)

    def _render(self, H, W, K, rays, near, far, constant_init):
        chunk = getattr(self.args, "chunk", 1024 * 32)
        direct = isinstance(rays, RB.RayColumns) and rays.shape[0] <= chunk \
            and (rays.viewdirs is not None) == bool(self.kw.get("use_viewdirs", False))
        if direct:     # the columns go straight into render_rays: no packing, no slicing
            if self.kw.get("ndc", True):      # forward-facing scenes: render's warp (run_plnerf.py:153-155) on the columns;
                # the view directions stay those of the camera-space rays (:146-150), near / far the caller's (0, 1)
                o_ndc, d_ndc = ndc_rays(H, W, K[0][0], 1., rays.rays_o, rays.rays_d)
                rays = RB.RayColumns(o_ndc, d_ndc, rays.near, rays.far, rays.viewdirs)
            kw = {k: v for k, v in self.kw.items() if k not in ("ndc", "use_viewdirs")}
            ret = render_rays(rays, retraw=True, constant_init=constant_init, **kw)
            return ret['rgb_map'], ret
        if isinstance(rays, RB.RayColumns):
            rays = (rays.rays_o, rays.rays_d)
        rgb, disp, acc, extras = render(H, W, K, chunk=chunk, rays=rays, near=near, far=far, retraw=True,
                                        constant_init=constant_init, **self.kw)
        return rgb, extras

    def _exchange_and_step(self, opts, exchange, guarded_by):
        """Finish the gradient exchange of the networks `exchange` (data parallel only) and step the optimizers `opts`,
        opts[k] guarded by the networks guarded_by[k].  With optim.FlatAdam the 1 / world factor and the guard -- the
        ranks' summed range status of those networks -- go into the step kernel: between the backward's last kernel and
        Adam there is one wait and no launch."""
        if self.bucket is None:
            for opt in opts:
                opt.step()
            return
        flat_adam = all(isinstance(opt, FlatAdam) and all(fl is not None for fl in opt._flat) for opt in opts)
        scale = self.bucket.finish(exchange, defer_scale=flat_adam)
        for opt, nets in zip(opts, guarded_by):
            if not flat_adam:
                opt.step()
                continue
            tails = self.bucket.tails(nets)
            opt.step(grad_scale=scale, guards=tails if len(tails) == len(nets) else None)

    def _step(self, H, W, K, rays, target_s, near, far):
        i = self.global_step + 1                      # the reference iterates i = start+1 .. N_iters
        n_local = rays.shape[0] if isinstance(rays, RB.RayColumns) else rays[0].reshape(-1, 3).shape[0]
        prev = Fn.DRAWS
        if self.draws is not None:
            self.draws.step, self.draws.ray_id0 = self.global_step, self.rank * n_local
            Fn.set_draw_source(self.draws)
        tape = Fn.MlpTape() if self.merged_backward else None
        Fn.MLP_TAPE = tape
        try:
            rgb, extras = self._render(H, W, K, rays, near, far, i < getattr(self.args, "constant_init", 0))
        finally:
            Fn.set_draw_source(prev)
            Fn.MLP_TAPE = None
        rgb0 = extras.get('rgb0')
        self.optimizer.zero_grad()
        self.optimizer_coarse.zero_grad()
        if rgb.is_cuda and rgb.dim() == 2 and rgb.shape == target_s.shape:
            # img2mse(rgb) + img2mse(rgb0), the psnr and both image gradients in one launch (:1287-1300);
            # backward((rgb, rgb0), (d loss / d rgb, d loss / d rgb0)) is loss.backward()
            loss4, g_rgb, g_rgb0 = Fn.image_loss_and_grads(rgb, rgb0, target_s)
            loss, psnr = loss4[0], loss4[3]
            if tape is not None and rgb0 is not None and merged_backward_ok(tape, self.nets):
                backward_merged(tape, self.nets, (rgb, rgb0), (g_rgb, g_rgb0), self.bucket)
                self.merged_steps += 1
            else:
                torch.autograd.backward((rgb,) if rgb0 is None else (rgb, rgb0),
                                        (g_rgb,) if rgb0 is None else (g_rgb, g_rgb0))
        else:
            img_loss = img2mse(rgb, target_s)
            loss = img_loss if rgb0 is None else img_loss + img2mse(rgb0, target_s)
            psnr = mse2psnr(img_loss.detach())
            loss.backward()
        # (fine optimizer first, as the reference does; with ONE network both optimizers step the same weights, :438-447)
        self._exchange_and_step([self.optimizer, self.optimizer_coarse], self.nets, [self.nets, self.nets[:1]])
        new_lrate = self.learning_rate()
        for group in self.optimizer.param_groups:
            group['lr'] = new_lrate
        for group in self.optimizer_coarse.param_groups:
            group['lr'] = new_lrate                   # sic: the reference uses the fine rate here (line 1315)
        self.global_step += 1
        if self.range_check_every and self.global_step % self.range_check_every == 0:
            self.check_range()
        return loss.detach(), psnr

    def check_range(self):
        """Look at the networks' range status words (one 4-byte read each).  Steps the guarded Adam kernels withheld are
        first taken back out of the optimizers' step counts; a set word raises FloatingPointError -- on every rank of a
        data-parallel job at the same step: the ranks' words travel with the gradients (dp.GradientBucket.tails), so a
        rank whose own words are clear still sees the withheld steps."""
        withheld = 0
        for opt in (self.optimizer, self.optimizer_coarse):
            if hasattr(opt, "withheld_steps"):
                withheld += opt.withheld_steps()
        for net in self.nets:
            if getattr(net, "precision", None) in L.GUARDED_PRECISIONS and net.is_supported() and \
                    next(net.parameters()).is_cuda:
                net.check_range()
        if withheld and self.bucket is not None:
            raise FloatingPointError(
                f"plnerf_amd: {withheld} optimizer step(s) were withheld because another rank's forward left the "
                "IEEE-half range (that rank's check_range() says which network); see NeRF.check_range for the remedy")


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
