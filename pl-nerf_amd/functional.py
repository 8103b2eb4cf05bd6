"""torch.autograd glue around the HIP kernels: one Function per differentiable stage.

PyTorch is plumbing here (device memory, streams, the autograd tape); every operation of
the path itself runs in libplnerf_hip.so.
"""
import ctypes

import numpy as np
import torch

from . import _lib as L


def _f32c(t):
    return t.detach().to(torch.float32).contiguous()


def _expect(cond, what):
    if not cond:
        raise ValueError(f"plnerf_amd: {what}")


class KernelTimer:
    """Optional per-launch timing with HIP events recorded on the launch stream (the stream
    the kernels are enqueued on is torch's current stream).  bench.py installs one to measure
    the dominant kernel inside the timed region; None (the default) costs nothing."""

    def __init__(self):
        self.events = {}

    def bracket(self, name):
        s = torch.cuda.Event(enable_timing=True)
        e = torch.cuda.Event(enable_timing=True)
        self.events.setdefault(name, []).append((s, e))
        return s, e

    def mean_ms(self, name, skip=0):
        ev = self.events.get(name, [])[skip:]
        if not ev:
            return None
        return sum(s.elapsed_time(e) for s, e in ev) / len(ev)

    def all_ms(self, name, skip=0):
        """Every bracket of `name` in milliseconds (call after a device synchronisation)."""
        return [s.elapsed_time(e) for s, e in self.events.get(name, [])[skip:]]

    def count(self, name):
        return len(self.events.get(name, []))


KERNEL_TIMER = None


GRAD_TAIL = 4      # floats behind a network's flat gradient (16 bytes: [0] = range status, the rest unused)


class QuadratureFn(torch.autograd.Function):
    """raw2outputs (run_plnerf.py:553-624) -> plnerf_quad_fwd / plnerf_quad_bwd.

    Differentiable with respect to `raw` through every output.  On the reference's NVS path the
    only consumer of tau and T (the sampler) is detached (run_plnerf.py:728) and they receive no
    gradient; the depth-supervised variant differentiates through the sampler
    (depth_supervised_exps/run_nerf_sample_based_depth.py:923-934), which is what their upstream
    gradients are for."""

    @staticmethod
    def forward(ctx, raw, z, near, far, rays_d, noise, mode, color_mode, white_bkgd, farcolorfix):
        _expect(z.dim() == 2, f"z_vals must be [rays, samples], got {tuple(z.shape)}")
        R, S = z.shape
        dev = raw.device
        _expect(tuple(raw.shape) == (R, S, 4), f"raw must be [{R}, {S}, 4] (rgb, sigma), got {tuple(raw.shape)}")
        _expect(near.numel() == R and far.numel() == R, f"near / far must hold one value per ray ({R})")
        _expect(tuple(rays_d.shape) == (R, 3), f"rays_d must be [{R}, 3], got {tuple(rays_d.shape)}")
        _expect(noise is None or tuple(noise.shape) == (R, S), f"noise must be [{R}, {S}]")
        raw_c, z_c = _f32c(raw), _f32c(z)
        near_c, far_c = _f32c(near).reshape(-1), _f32c(far).reshape(-1)
        d_c = _f32c(rays_d)
        noise_c = None if noise is None else _f32c(noise)
        linear = mode == "linear"
        n = S + 1 if linear else S
        rgb = torch.empty(R, 3, device=dev)
        disp = torch.empty(R, device=dev)
        acc = torch.empty(R, device=dev)
        depth = torch.empty(R, device=dev)
        w = torch.empty(R, n, device=dev)
        tau = torch.empty(R, S + 2, device=dev) if linear else torch.empty(0, device=dev)
        T = torch.empty(R, S + 2, device=dev) if linear else torch.empty(0, device=dev)
        if R > 0:
            L.check(L.lib().plnerf_quad_fwd(
              L.dptr(raw_c, "raw"), L.dptr(z_c, "z_vals"), L.dptr(near_c, "near"), L.dptr(far_c, "far"),
              L.dptr(d_c, "rays_d"), L.dptr(noise_c, "noise"), R, S, L.MODE[mode], L.COLOR[color_mode],
              int(bool(white_bkgd)), int(bool(farcolorfix)), L.dptr(rgb), L.dptr(disp), L.dptr(acc),
              L.dptr(depth), L.dptr(w), L.dptr(tau) if linear else None, L.dptr(T) if linear else None,
              L.stream()), "plnerf_quad_fwd")
        ctx.save_for_backward(raw_c, z_c, near_c, far_c, d_c, noise_c if noise_c is not None else torch.empty(0),
                              depth, acc)
        ctx.cfg = (mode, color_mode, bool(white_bkgd), bool(farcolorfix), noise_c is not None)
        ctx.near_shape, ctx.far_shape = near.shape, far.shape
        if not linear:
            ctx.mark_non_differentiable(tau, T)
        ctx.set_materialize_grads(False)
        return rgb, disp, acc, w, depth, tau, T

    @staticmethod
    def backward(ctx, g_rgb, g_disp, g_acc, g_w, g_depth, g_tau, g_T):
        if not any(ctx.needs_input_grad[1:5]):
            g_raw = _quad_backward(ctx.saved_tensors, ctx.cfg, g_rgb, g_disp, g_acc, g_w, g_depth, g_tau, g_T)
            return g_raw, None, None, None, None, None, None, None, None, None
        # the ray geometry asks too (z_vals / near / far / rays_d came from a ray batch that requires a gradient:
        # run_plnerf.py:707, 735 under autograd) -- plnerf_quad_bwd_rays
        g_raw, g_z, g_near, g_far, g_dnorm = _quad_backward(ctx.saved_tensors, ctx.cfg, g_rgb, g_disp, g_acc, g_w, g_depth,
                                                            g_tau, g_T, geometry=True)
        d_c = ctx.saved_tensors[4]
        need = ctx.needs_input_grad
        g_d = (g_dnorm[:, None] * (d_c / torch.norm(d_c, dim=-1, keepdim=True))) if need[4] else None
        return (g_raw, g_z if need[1] else None, g_near.reshape(ctx.near_shape) if need[2] else None,
                g_far.reshape(ctx.far_shape) if need[3] else None, g_d, None, None, None, None, None)


# Armed (a list) by a caller that runs plnerf_quad_bwd and the MLP backward itself, back to back (train.TrainStep's merged
# backward): every plnerf_quad_bwd then leaves its by-product and logs (g_raw, its version counter, the int32 tensor of maxima)
# here, and the caller hands the matching tensors to mlp_backward_multi.  None: no by-product is asked for.  (The by-product: one
# uint32 per workgroup of the launch = the fp32 bits of the largest |g_raw| among its rays; plnerf_mlp_bwd takes the array
# as `g_absmax` and skips its own pass over g_raw and its memset.)
ABSMAX_LOG = None


def _quad_backward(saved, cfg, g_rgb, g_disp, g_acc, g_w, g_depth, g_tau, g_T, geometry=False):
    """plnerf_quad_bwd from the tensors a quadrature forward saved (shared by QuadratureFn and CoarseEpilogueFn).  With
    ABSMAX_LOG armed the launch also leaves its workgroups' max |g_raw| in a tensor and logs it.  geometry=True:
    plnerf_quad_bwd_rays -- returns (g_raw, g_z, g_near, g_far, g_dnorm)."""
    raw_c, z_c, near_c, far_c, d_c, noise_c, depth, acc = saved
    mode, color_mode, white_bkgd, farcolorfix, has_noise = cfg
    R, S = z_c.shape
    dev = raw_c.device
    g_rgb = torch.zeros(R, 3, device=dev) if g_rgb is None else _f32c(g_rgb)
    g_depth = None if g_depth is None else _f32c(g_depth)
    g_acc = None if g_acc is None else _f32c(g_acc)
    if g_disp is not None:
        # disp = 1 / max(1e-10, depth/acc)  (run_plnerf.py:617): fold into depth / acc
        ratio = depth / acc
        live = (ratio > 1e-10).to(torch.float32) * _f32c(g_disp)
        gd = -live * acc / (depth * depth)
        ga = live / depth
        gd = torch.where(torch.isfinite(gd), gd, torch.zeros_like(gd))
        ga = torch.where(torch.isfinite(ga), ga, torch.zeros_like(ga))
        g_depth = gd if g_depth is None else g_depth + gd
        g_acc = ga if g_acc is None else g_acc + ga
    g_w = None if g_w is None else _f32c(g_w)
    g_tau = None if (g_tau is None or mode != "linear") else _f32c(g_tau)
    g_T = None if (g_T is None or mode != "linear") else _f32c(g_T)
    g_raw = torch.empty(R, S, 4, device=dev)
    if geometry:
        g_z = torch.empty(R, S, device=dev)
        g_near, g_far, g_dnorm = (torch.empty(R, device=dev) for _ in range(3))
        if R > 0:
            L.check(L.lib().plnerf_quad_bwd_rays(
              L.dptr(raw_c), L.dptr(z_c), L.dptr(near_c), L.dptr(far_c), L.dptr(d_c),
              L.dptr(noise_c) if has_noise else None, R, S, L.MODE[mode], L.COLOR[color_mode],
              int(white_bkgd), int(farcolorfix), L.dptr(g_rgb), L.dptr(g_depth), L.dptr(g_acc), L.dptr(g_w),
              L.dptr(g_tau), L.dptr(g_T), L.dptr(g_raw), L.dptr(g_z), L.dptr(g_near), L.dptr(g_far), L.dptr(g_dnorm),
              L.stream()), "plnerf_quad_bwd_rays")
        return g_raw, g_z, g_near, g_far, g_dnorm
    if R > 0:
        cand = None
        if ABSMAX_LOG is not None:
            cand = torch.empty((R + L.QUAD_RAYS_PER_GROUP - 1) // L.QUAD_RAYS_PER_GROUP, device=dev, dtype=torch.int32)
        L.check(L.lib().plnerf_quad_bwd(
          L.dptr(raw_c), L.dptr(z_c), L.dptr(near_c), L.dptr(far_c), L.dptr(d_c),
          L.dptr(noise_c) if has_noise else None, R, S, L.MODE[mode], L.COLOR[color_mode],
          int(white_bkgd), int(farcolorfix), L.dptr(g_rgb), L.dptr(g_depth), L.dptr(g_acc), L.dptr(g_w),
          L.dptr(g_tau), L.dptr(g_T), L.dptr(g_raw), L.dptr(cand, "absmax_out", torch.int32), L.stream()), "plnerf_quad_bwd")
        if cand is not None:
            ABSMAX_LOG.append((g_raw, g_raw._version, cand))
    return g_raw


class DrawSource:
    """Counter-based uniform draws (csrc/philox.h): a draw is a function of (seed, step, which draw of the step, the
    ray's GLOBAL id, column) only, so a batch sharded over N ranks (ray_id0 = the rank's first global ray) sees the
    numbers one rank would see.  Install with `set_draw_source`; render_rays then takes its stratified jitter and
    its sampler draws from here (in the consuming kernels themselves on the fused path) instead of torch.rand."""
    T_RAND, U, NOISE = 0, 1, 2          # stream ids (the density noise takes NOISE for the coarse pass, NOISE + 1 for the fine one)

    def __init__(self, seed=0, ray_id0=0, step=0):
        self.seed, self.ray_id0, self.step = int(seed), int(ray_id0), int(step)
        self.chunk_offset = 0
        self.noise_calls = 0      # raw2outputs calls of the current render_rays (coarse pass, then fine pass)

    def first_ray(self):
        return self.ray_id0 + self.chunk_offset

    def uniform(self, R, n, stream_id, device):
        out = torch.empty(R, n, device=device)
        if R > 0:
            L.check(L.lib().plnerf_uniform(self.seed, stream_id, self.step, self.first_ray(), R, n, L.dptr(out),
                                           L.stream()), "plnerf_uniform")
        return out

    def normal(self, R, n, stream_id, device):
        """Standard normal draws [R, n] from the same counters (plnerf_normal)."""
        out = torch.empty(R, n, device=device)
        if R > 0:
            L.check(L.lib().plnerf_normal(self.seed, stream_id, self.step, self.first_ray(), R, n, L.dptr(out),
                                          L.stream()), "plnerf_normal")
        return out

    def next_noise_stream(self):
        """Stream id of this raw2outputs call's density noise: the coarse and the fine pass of a render_rays call (of a
        chunk of it) alternate; the ray's global id does the rest."""
        sid = self.NOISE + (self.noise_calls & 1)
        self.noise_calls += 1
        return sid


DRAWS = None


def set_draw_source(src):
    """Install (or with None remove) the process-wide DrawSource; returns the previous one."""
    global DRAWS
    prev, DRAWS = DRAWS, src
    return prev


class CoarseEpilogueFn(torch.autograd.Function):
    """run_plnerf.py:714-735 in piecewise-linear mode as one launch (plnerf_coarse_epilogue): raw2outputs of the
    coarse pass, sample_pdf_reformulation, clamp, sort(cat), the fine pass's sample positions and z_std.
    Differentiable with respect to `raw` through the coarse maps (backward = plnerf_quad_bwd); the samples are
    detached on the reference path (:728).  `u` None = drawn inside the kernel from `draws`."""

    @staticmethod
    def forward(ctx, raw, z, near, far, rays_o, rays_d, noise, u, N, color_mode, white_bkgd, farcolorfix, zero_tol,
                eps, draws, want_weights=False):
        R, S = z.shape
        dev = raw.device
        _expect(tuple(raw.shape) == (R, S, 4), f"raw must be [{R}, {S}, 4], got {tuple(raw.shape)}")
        raw_c, z_c = _f32c(raw), _f32c(z)
        near_c, far_c = _f32c(near).reshape(-1), _f32c(far).reshape(-1)
        o_c, d_c = _f32c(rays_o), _f32c(rays_d)
        noise_c = None if noise is None else _f32c(noise)
        u_c = None if u is None else _f32c(u)
        _expect(u_c is None or tuple(u_c.shape) in ((N,), (R, N)), "u must be [N] or [R, N]")
        _expect(u_c is not None or draws is not None, "no draws: pass u or a DrawSource")
        stride = 0 if (u_c is None or u_c.dim() == 1) else N
        rgb = torch.empty(R, 3, device=dev)
        disp, acc, depth, z_std = (torch.empty(R, device=dev) for _ in range(4))
        z_fine = torch.empty(R, S + N, device=dev)
        pts = torch.empty(R, S + N, 3, device=dev)
        # (the coarse weights reach HBM only for a caller that returns them: the depth-supervised variant's `weights0`)
        weights = torch.empty(R, S + 1, device=dev) if want_weights else None
        seed, step, ray0 = (draws.seed, draws.step, draws.first_ray()) if draws is not None else (0, 0, 0)
        L.check(L.lib().plnerf_coarse_epilogue(
            L.dptr(raw_c, "raw"), L.dptr(z_c, "z_vals"), L.dptr(near_c, "near"), L.dptr(far_c, "far"),
            L.dptr(o_c, "rays_o"), L.dptr(d_c, "rays_d"), L.dptr(noise_c, "noise"), L.dptr(u_c, "u"), stride, seed, step,
            ray0, R, S, int(N), L.COLOR[color_mode], int(bool(white_bkgd)), int(bool(farcolorfix)), float(zero_tol),
            float(eps), L.dptr(rgb), L.dptr(disp), L.dptr(acc), L.dptr(depth), L.dptr(weights), None, None, L.dptr(z_fine),
            L.dptr(pts), L.dptr(z_std), L.stream()), "plnerf_coarse_epilogue")
        ctx.save_for_backward(raw_c, z_c, near_c, far_c, d_c, noise_c if noise_c is not None else torch.empty(0),
                              depth, acc)
        ctx.cfg = ("linear", color_mode, bool(white_bkgd), bool(farcolorfix), noise_c is not None)
        if weights is None:
            weights = torch.empty(0, device=dev)
        ctx.mark_non_differentiable(z_fine, pts, z_std, weights)
        ctx.set_materialize_grads(False)
        if want_weights:
            return rgb, disp, acc, depth, z_fine, pts, z_std, weights
        return rgb, disp, acc, depth, z_fine, pts, z_std

    @staticmethod
    def backward(ctx, g_rgb, g_disp, g_acc, g_depth, g_z, g_pts, g_std, g_w=None):
        g_raw = _quad_backward(ctx.saved_tensors, ctx.cfg, g_rgb, g_disp, g_acc, None, g_depth, None, None)
        return (g_raw,) + (None,) * 15


class FineEpilogueFn(torch.autograd.Function):
    """The depth-supervised variant's last stage in piecewise-linear mode as one launch (plnerf_fine_epilogue):
    raw2outputs of the final pass and sample_pdf_reformulation_return_u on its weights / tau / T -> the depth
    hypotheses (depth_supervised_exps/run_nerf_sample_based_depth.py:909-934), plus z_std = std(hypotheses).
    Differentiable with respect to `raw` through the maps, the weights AND the hypotheses (which keep their tape in the
    reference: :923-934): backward = plnerf_sample_pl_bwd (g_hyp -> g_tau, g_T), then plnerf_quad_bwd.
    u: [R, N], one shared row [N] (is_joint), or None = drawn in the kernel from `draws`."""
    HYP_STREAM = 4

    @staticmethod
    def forward(ctx, raw, z, near, far, rays_d, noise, u, N, color_mode, white_bkgd, farcolorfix, zero_tol, eps, draws):
        R, S = z.shape
        dev = raw.device
        _expect(tuple(raw.shape) == (R, S, 4), f"raw must be [{R}, {S}, 4], got {tuple(raw.shape)}")
        raw_c, z_c = _f32c(raw), _f32c(z)
        near_c, far_c = _f32c(near).reshape(-1), _f32c(far).reshape(-1)
        d_c = _f32c(rays_d)
        noise_c = None if noise is None else _f32c(noise)
        u_c = None if u is None else _f32c(u)
        _expect(u_c is None or tuple(u_c.shape) in ((N,), (R, N)), "u must be [N] or [R, N]")
        _expect(u_c is not None or draws is not None, "no draws: pass u or a DrawSource")
        stride = 0 if (u_c is None or u_c.dim() == 1) else N
        rgb = torch.empty(R, 3, device=dev)
        disp, acc, depth, z_std = (torch.empty(R, device=dev) for _ in range(4))
        w = torch.empty(R, S + 1, device=dev)
        tau, T = torch.empty(R, S + 2, device=dev), torch.empty(R, S + 2, device=dev)
        hyp = torch.empty(R, N, device=dev)
        inds = torch.empty(R, N, device=dev, dtype=torch.int64)
        u_used = u_c if (u_c is not None and u_c.dim() == 2) else torch.empty(R, N, device=dev)
        seed, step, ray0 = (draws.seed, draws.step, draws.first_ray()) if draws is not None else (0, 0, 0)
        L.check(L.lib().plnerf_fine_epilogue(
            L.dptr(raw_c, "raw"), L.dptr(z_c, "z_vals"), L.dptr(near_c, "near"), L.dptr(far_c, "far"), L.dptr(d_c, "rays_d"),
            L.dptr(noise_c, "noise"), L.dptr(u_c, "u"), stride, seed, step, ray0, R, S, int(N), L.COLOR[color_mode],
            int(bool(white_bkgd)), int(bool(farcolorfix)), float(zero_tol), float(eps), L.dptr(rgb), L.dptr(disp),
            L.dptr(acc), L.dptr(depth), L.dptr(w), L.dptr(tau), L.dptr(T), L.dptr(hyp), L.dptr(inds, "inds", torch.int64),
            None if u_used is u_c else L.dptr(u_used), L.dptr(z_std), L.stream()), "plnerf_fine_epilogue")
        ctx.save_for_backward(raw_c, z_c, near_c, far_c, d_c, noise_c if noise_c is not None else torch.empty(0),
                              depth, acc, tau, T, u_used, inds)
        ctx.cfg = ("linear", color_mode, bool(white_bkgd), bool(farcolorfix), noise_c is not None)
        ctx.sampler = (float(zero_tol), float(eps))
        ctx.mark_non_differentiable(u_used, inds, z_std)
        ctx.set_materialize_grads(False)
        return rgb, disp, acc, depth, w, tau, T, hyp, u_used, inds, z_std

    @staticmethod
    def backward(ctx, g_rgb, g_disp, g_acc, g_depth, g_w, g_tau, g_T, g_hyp, g_u, g_inds, g_std):
        raw_c, z_c, near_c, far_c, d_c, noise_c, depth, acc, tau, T, u_used, inds = ctx.saved_tensors
        if g_hyp is not None:
            R, S = z_c.shape
            N = inds.shape[-1]
            zero_tol, eps = ctx.sampler
            gt, gT = torch.empty(R, S + 2, device=z_c.device), torch.empty(R, S + 2, device=z_c.device)
            L.check(L.lib().plnerf_sample_pl_bwd(
                L.dptr(z_c), L.dptr(tau), L.dptr(T), L.dptr(near_c), L.dptr(far_c), L.dptr(u_used), N,
                L.dptr(inds, "inds", torch.int64), L.dptr(_f32c(g_hyp)), R, S, N, zero_tol, eps, L.dptr(gt), L.dptr(gT),
                L.stream()), "plnerf_sample_pl_bwd")
            g_tau = gt if g_tau is None else g_tau + gt
            g_T = gT if g_T is None else g_T + gT
        g_raw = _quad_backward((raw_c, z_c, near_c, far_c, d_c, noise_c, depth, acc), ctx.cfg, g_rgb, g_disp, g_acc, g_w,
                               g_depth, g_tau, g_T)
        return (g_raw,) + (None,) * 13


_LOSS_WS = {}


def _loss_workspace(device, nbytes, kernel):
    """A loss kernel's partial sums and ticket counter: zeroed once per (kernel, device, stream), left zeroed by every
    launch of THAT kernel (each lays its partials and its ticket out differently; two streams may run one at the same
    time: one workspace each)."""
    key = (kernel, device, L.stream().value, nbytes)
    ws = _LOSS_WS.get(key)
    if ws is None:
        ws = _LOSS_WS[key] = torch.zeros(nbytes // 8, device=device, dtype=torch.float64)
    return ws


def _image_loss(rgb, rgb0, target, coarse_loss=None):
    rgb_c, t_c = _f32c(rgb), _f32c(target)
    rgb0_c = None if rgb0 is None else _f32c(rgb0)
    _expect(rgb_c.shape == t_c.shape and rgb_c.dim() == 2 and rgb_c.shape[1] == 3, "rgb / target must be [R, 3]")
    _expect(rgb0_c is None or coarse_loss is None, "coarse_loss replaces rgb0")
    loss4 = torch.empty(4, device=rgb_c.device)
    g1 = torch.empty_like(rgb_c)
    g0 = None if rgb0_c is None else torch.empty_like(rgb_c)
    ws = _loss_workspace(rgb_c.device, L.IMAGE_LOSS_WORKSPACE_BYTES, "image")
    L.check(L.lib().plnerf_image_loss(L.dptr(rgb_c, "rgb"), L.dptr(rgb0_c, "rgb0"), L.dptr(t_c, "target"),
                                      rgb_c.shape[0], L.dptr(loss4), L.dptr(g1), L.dptr(g0), L.dptr(coarse_loss, "coarse_loss"),
                                      L.dptr(ws, "workspace", torch.float64), L.stream()), "plnerf_image_loss")
    return loss4, g1, g0


class ImageLossFn(torch.autograd.Function):
    """img2mse(rgb, target) + img2mse(rgb0, target) (run_plnerf.py:1287-1296) -> (total, fine, coarse), one launch
    that also leaves d total / d rgb and d total / d rgb0; the backward scales them by the upstream gradients."""

    @staticmethod
    def forward(ctx, rgb, rgb0, target):
        loss4, g1, g0 = _image_loss(rgb, rgb0, target)
        ctx.grads = (g1, g0)
        ctx.set_materialize_grads(False)
        total, fine, coarse, _ = loss4.unbind(0)
        return total, fine, coarse

    @staticmethod
    def backward(ctx, g_total, g_fine, g_coarse):
        g1, g0 = ctx.grads

        def scaled(g, *ups):
            ups = [x for x in ups if x is not None]
            if g is None or not ups:
                return None
            w = ups[0] if len(ups) == 1 else ups[0] + ups[1]
            return g * w
        return scaled(g1, g_total, g_fine), scaled(g0, g_total, g_coarse), None


def image_loss_and_grads(rgb, rgb0, target, coarse_loss=None):
    """plnerf_image_loss without the autograd wrapper, for a caller that back-propagates the two gradients itself
    (train.TrainStep: torch.autograd.backward((rgb, rgb0), (g_rgb, g_rgb0)) is loss.backward() minus three tiny
    launches).  Returns (loss4 = [total, fine, coarse, psnr], g_rgb, g_rgb0).  `coarse_loss` (with rgb0 None): the loss4
    of an earlier call on the coarse image alone, whose [1] becomes this call's coarse term (a caller that starts the
    coarse network's backward before the fine pass exists still gets the reference's total)."""
    return _image_loss(rgb, rgb0, target, coarse_loss)


def coarse_samples(rays_o, rays_d, near, far, t_vals, t_rand, lindisp, perturb, draws):
    """plnerf_coarse_samples: (z_vals [R,S], pts [R,S,3]) of run_plnerf.py:683-708.  With perturb and t_rand None
    the jitter is drawn in the kernel from `draws`."""
    o_c, d_c = _f32c(rays_o), _f32c(rays_d)
    near_c, far_c = _f32c(near).reshape(-1), _f32c(far).reshape(-1)
    t_c = _f32c(t_vals)
    r_c = _f32c(t_rand) if t_rand is not None else None
    R, S = near_c.shape[0], t_c.shape[0]
    _expect(not perturb or r_c is not None or draws is not None, "perturb needs t_rand or a DrawSource")
    z = torch.empty(R, S, device=near_c.device)
    pts = torch.empty(R, S, 3, device=near_c.device)
    seed, step, ray0 = (draws.seed, draws.step, draws.first_ray()) if draws is not None else (0, 0, 0)
    L.check(L.lib().plnerf_coarse_samples(
        L.dptr(o_c, "rays_o"), L.dptr(d_c, "rays_d"), L.dptr(near_c, "near"), L.dptr(far_c, "far"), L.dptr(t_c, "t_vals"),
        L.dptr(r_c, "t_rand"), seed, step, ray0, R, S, int(bool(lindisp)), int(bool(perturb)), L.dptr(z), L.dptr(pts),
        L.stream()), "plnerf_coarse_samples")
    return z, pts


def _grad_buffer(ctx, dev, zero=False, block=None):
    """A network's 24 gradients as consecutive slices of ONE buffer, in parameter order: optim.FlatAdam and
    dp.GradientBucket then see the network's gradient as a single flat tensor (one Adam launch, one all-reduce without
    gather / scatter copies).  GRAD_TAIL floats behind them: [0] = the network's range status as the reduction kernel
    leaves it, so that a data-parallel exchange of this buffer carries it along (dp.GradientBucket)."""
    sizes = [int(torch.Size(sh).numel()) for sh in ctx.param_shapes]
    n_grad = sum(sizes)
    # (`block`: a slice of a buffer that holds several networks' gradients back to back -- one collective for all of them)
    full = (torch.zeros if zero else torch.empty)(n_grad + GRAD_TAIL, device=dev, dtype=torch.float32) if block is None else block
    if ctx.net.__dict__.get("_wants_grad_flat"):      # (a dp.GradientBucket is attached: it exchanges the buffer WITH its tail,
        ctx.net.__dict__["_grad_flat"] = full         #  and lets go of this reference once it has -- GradientBucket.finish)
    return [t.view(sh) for t, sh in zip(full[:n_grad].split(sizes), ctx.param_shapes)], full


def _mlp_backward_launch(ctxs, g_raws, absmax_log=None):
    """plnerf_mlp_bwd_multi over the saved state of one or two MlpFn forwards (the same precision, input widths and
    density activation): one launch sequence for all of them.  absmax_log: [(g_raw, version, maxima tensor)] of the
    plnerf_quad_bwd launches that produced these g_raws in THIS backward pass (ABSMAX_LOG), or None.  Returns ([the 24
    gradient views per job], [workspace per job]); with a dp.GradientBucket attached, each job's flat buffer is left on its network as `_grad_flat` until the exchange."""
    n = len(ctxs)
    c0 = ctxs[0]
    dev = g_raws[0].device
    _expect(all(c.prec == c0.prec and c.beta == c0.beta and int(c.net.input_ch) == int(c0.net.input_ch) and
                int(c.net.hip_view_ch) == int(c0.net.hip_view_ch) for c in ctxs),
            "jobs of one backward launch share precision, input widths and density activation")
    gs, wss, grads_all, fulls, absmax = [], [], [], [], []
    # the jobs' gradient buffers back to back in ONE allocation: a data-parallel exchange of a merged backward is then one
    # all-reduce over both networks (dp.GradientBucket.gradients_ready) instead of two latency-bound ones
    sizes = [sum(int(torch.Size(sh).numel()) for sh in c.param_shapes) + GRAD_TAIL for c in ctxs]
    block = torch.empty(sum(sizes), device=dev, dtype=torch.float32) if n > 1 else None
    offs = [sum(sizes[:j]) for j in range(n)]
    for j, (c, g_raw) in enumerate(zip(ctxs, g_raws)):
        g = _f32c(g_raw)
        gs.append(g)
        # (the producer's by-product: the word of the plnerf_quad_bwd launch that wrote exactly this buffer -- and nothing has
        # written to it since: if `raw` ever gets a second consumer, autograd accumulates the second gradient INTO the first
        # one's buffer (same pointer, version counter bumped), and the maximum of the first alone would understate the scale
        # of the sum; the absmax pass then runs instead.  ADVICE r05)
        am = next((v for t, ver, v in (absmax_log or ())
                   if t.data_ptr() == g.data_ptr() and t.numel() == g.numel() and t._version == ver), None)
        absmax.append(am if (am is not None and c.beta == 0.0 and g.numel() == 4 * c.n_rows) else None)
        wss.append(torch.empty(L.lib().plnerf_mlp_bwd_workspace_bytes(c.n_rows, c.prec) // 4, device=dev, dtype=torch.float32))
        grads, full = _grad_buffer(c, dev, block=None if block is None else block[offs[j]:offs[j] + sizes[j]])
        grads_all.append(grads)
        fulls.append(full)
    vp = lambda items: (ctypes.c_void_p * n)(*[None if x is None else x.value for x in items])
    n_grad = [full.numel() - GRAD_TAIL for full in fulls]
    L.check(L.lib().plnerf_mlp_bwd_multi(
        n, vp([L.dptr(c.packed) for c in ctxs]), c0.prec, vp([L.dptr(g, "g_raw") for g in gs]),
        vp([L.dptr(a, "g_absmax", torch.int32) for a in absmax]), (ctypes.c_int * n)(*[0 if a is None else a.numel() for a in absmax]),
        int(c0.net.input_ch), int(c0.net.hip_view_ch),
        (ctypes.c_int * n)(*[c.n_rows for c in ctxs]), vp([L.dptr(c.saved_acts) for c in ctxs]),
        (ctypes.c_int * n)(*[c.saved_layout for c in ctxs]),
        vp([L.dptr(c.saved_tensors[0]) if c.beta > 0.0 else None for c in ctxs]), c0.beta, vp([L.dptr(w) for w in wss]),
        L.ptr_table([t for grads in grads_all for t in grads], "grads"),
        vp([ctypes.c_void_p(full.data_ptr() + 4 * k) for full, k in zip(fulls, n_grad)]), L.stream()),
        "plnerf_mlp_bwd_multi")
    global ABSMAX_HITS
    ABSMAX_HITS += sum(a is not None for a in absmax)
    return grads_all, wss


ABSMAX_HITS = 0      # (diagnostics / tests: backward jobs that took max |g_raw| from plnerf_quad_bwd's by-product)


class MlpTape:
    """Armed by train.TrainStep around a render: records the output tensor of every MlpFn forward (NeRF.query /
    NeRF.forward append to it), so that the step can run the two networks' backward as ONE launch sequence
    (mlp_backward_multi) instead of letting autograd run them one after the other."""

    def __init__(self):
        self.outs = []


MLP_TAPE = None


def mlp_backward_multi(outs, g_raws, absmax_log=None):
    """The backward of several MlpFn forwards at once -- `outs`: their output tensors (each one's grad_fn holds the saved
    state), `g_raws`: the upstream gradients -- through plnerf_mlp_bwd_multi: one dgrad grid and one launch of each
    weight-gradient kernel for all of them.  Only for networks whose kernel parameters ARE their nn.Parameters
    (NeRF.is_native()) and whose inputs / camera code take no gradient; returns the per-network lists of 24 gradient
    tensors (slices of each network's flat buffer), which the caller assigns to `.grad`."""
    ctxs = [o.grad_fn for o in outs]
    for c in ctxs:
        _expect(getattr(c, "saved_acts", None) is not None and c.n_rows > 0 and not c.in_grad and not c.n_cam,
                "mlp_backward_multi: a forward without saved state, or one whose inputs take a gradient")
    timer = KERNEL_TIMER
    if timer is not None:
        ev = timer.bracket("mlp_bwd[" + "+".join(str(c.n_rows) for c in ctxs) + "]")
        ev[0].record()
    grads_all, _ = _mlp_backward_launch(ctxs, g_raws, absmax_log)
    if timer is not None:
        ev[1].record()
    for c in ctxs:
        c.saved_acts = None
    return grads_all


class MlpFn(torch.autograd.Function):
    """Embedder + NeRF.forward (run_nerf_helpers.py:24-54, 105-128) -> plnerf_mlp_fwd /
    plnerf_mlp_bwd (+ plnerf_mlp_input_grad).  Gradients flow to the 24 parameter tensors, to the inputs when they ask
    for one (pts / viewdirs through the in-kernel encoding's derivative, or the embedded rows; no reference training path
    does: the sample positions are detached there) and to `cam`: the depth-supervised script's per-image camera code, a vector the caller has repeated into
    the LAST cam.numel() columns of every row of `embedded` (run_nerf_sample_based_depth.py:60-64, trained through the
    network input at :1091-1093, 1122-1123 and optimised alone at :311-345).  Because every row carries the same
    values, its gradient is the view layer's weight columns applied to the row-sum of dz_view -- which is the view
    layer's bias gradient, already produced by the backward:  g_cam = W_view[:, -n_cam:]^T  g_bias_view."""

    @staticmethod
    def forward(ctx, pts, viewdirs, embedded, cam, spr, net, want_grad, *params):
        pe_scale = float(getattr(net, "_query_scale", 1.0))      # (set by NeRF.query for the duration of the call)
        # the density activation (depth-supervised variant: softplus, beta 10) in the kernel's last store; its
        # derivative on the backward's entry, from the activated output saved below
        beta = float(getattr(net, "density_beta", 0.0))
        prec = L.PRECISION[net.precision]
        # A gradient for an MLP *input* (pts / viewdirs, or the embedded rows) is off every reference training path (the
        # samples are detached, run_plnerf.py:728) but is what autograd gives the reference module: plnerf_mlp_input_grad
        # after the backward's dgrad, then the encoding's own derivative (backward below).
        in_grad = bool(want_grad) and any(ctx.needs_input_grad[0:3])
        _expect(cam is None or (embedded is not None and cam.dim() == 1 and 0 < cam.numel() <= embedded.shape[-1]),
                "cam must be a vector held in the last columns of `embedded`")
        packed = net.packed_weights()
        if embedded is not None:
            emb_c = _f32c(embedded)
            n_rows, dev = emb_c.shape[0], emb_c.device
            pts_c = vd_c = None
        else:
            pts_c, vd_c = _f32c(pts), _f32c(viewdirs)
            n_rows, dev = pts_c.shape[0], pts_c.device
            emb_c = None
        raw = torch.empty(n_rows, 4, device=dev)
        # grad mode is always off inside Function.forward: the caller samples torch.is_grad_enabled()
        need_grad = in_grad or (bool(want_grad) and any(ctx.needs_input_grad[3:4] + ctx.needs_input_grad[7:]))
        saved = None
        if need_grad and n_rows > 0:
            nbytes = L.lib().plnerf_mlp_saved_bytes(n_rows, prec)
            saved = torch.empty(nbytes // 4, device=dev, dtype=torch.float32)
        timer = KERNEL_TIMER
        if timer is not None:
            ev = timer.bracket(f"mlp_fwd[{n_rows}]")
            ev[0].record()
        L.check(L.lib().plnerf_mlp_fwd(
            L.dptr(packed, "packed"), prec, L.dptr(pts_c, "pts"), L.dptr(vd_c, "viewdirs"),
            L.dptr(emb_c, "embedded"), int(net.input_ch), int(net.hip_view_ch), n_rows, int(spr), pe_scale, beta,
            L.dptr(raw), L.dptr(saved), L.FWD_KERNEL, L.stream()), "plnerf_mlp_fwd")
        if timer is not None:
            ev[1].record()
        ctx.net, ctx.prec, ctx.n_rows = net, prec, n_rows
        ctx.saved_layout = L.lib().plnerf_mlp_saved_layout(prec, int(emb_c is not None), L.FWD_KERNEL)
        ctx.saved_acts = saved
        ctx.packed = packed
        ctx.param_shapes = [p.shape for p in params]
        ctx.n_cam = 0 if cam is None else int(cam.numel())
        # (the view layer's weight, for the camera code's gradient: the packed copy is not in [out][in] order)
        ctx.view_weight = params[16].detach() if (ctx.n_cam and need_grad) else None
        ctx.beta = beta
        ctx.in_grad = in_grad and n_rows > 0
        if ctx.in_grad:      # (what the encoding's derivative needs; the 24 tensors the input-gradient kernel reads)
            ctx.in_pts, ctx.in_vd, ctx.in_spr, ctx.in_scale = pts_c, vd_c, int(spr), pe_scale
            ctx.in_params = [p.detach() for p in params]
            ctx.in_emb_shape = None if embedded is None else embedded.shape
        if beta > 0.0 and saved is not None:
            ctx.save_for_backward(raw)      # (an output: autograd keeps it without a reference cycle)
        return raw

    @staticmethod
    def backward(ctx, g_raw):
        n_rows = ctx.n_rows
        dev = g_raw.device
        if ctx.saved_acts is None and n_rows > 0:
            raise RuntimeError("plnerf_amd: backward through an MLP forward that ran without saved state")
        if n_rows == 0:
            grads, _ = _grad_buffer(ctx, dev, zero=True)
            return (None,) * 3 + (None if not ctx.n_cam else grads[0].new_zeros(ctx.n_cam),) + (None,) * 3 + tuple(grads)
        timer = KERNEL_TIMER
        if timer is not None:
            ev = timer.bracket(f"mlp_bwd[{n_rows}]")
            ev[0].record()
        (grads,), (ws,) = _mlp_backward_launch([ctx], [g_raw])
        if timer is not None:
            ev[1].record()
        g_cam = None
        if ctx.n_cam and ctx.view_weight is not None:
            # params: ..., views_linears.0.weight (16) [W/2, W + view_ch], views_linears.0.bias (17) [W/2]
            g_cam = torch.mv(ctx.view_weight[:, -ctx.n_cam:].t(), grads[17])
        g_pts = g_vd = g_embedded = None
        if ctx.in_grad:
            xyz_ch, dir_ch = int(ctx.net.input_ch), int(ctx.net.hip_view_ch)
            g_rows = torch.empty(n_rows, xyz_ch + dir_ch, device=dev, dtype=torch.float32)
            L.check(L.lib().plnerf_mlp_input_grad(L.ptr_table(ctx.in_params, "params"), ctx.prec, xyz_ch, dir_ch, n_rows,
                                                  L.dptr(ws), L.dptr(g_rows), L.stream()), "plnerf_mlp_input_grad")
            if ctx.in_emb_shape is not None:
                g_embedded = g_rows.view(ctx.in_emb_shape) if ctx.needs_input_grad[2] else None
            else:
                # the in-kernel encoding gamma(x) = [x, sin((x s) 2^k), cos((x s) 2^k)]_k (run_nerf_helpers.py:24-54; s =
                # the call's input scale): its transposed Jacobian applied to the rows' gradient
                def encoding_vjp(x, g, scale):
                    out = g[:, 0:3].clone()
                    for k in range((g.shape[1] - 3) // 6):
                        f = 2.0 ** k
                        arg = (x * scale) * f
                        out += (scale * f) * (torch.cos(arg) * g[:, 3 + 6 * k:6 + 6 * k]
                                              - torch.sin(arg) * g[:, 6 + 6 * k:9 + 6 * k])
                    return out
                if ctx.needs_input_grad[0]:
                    g_pts = encoding_vjp(ctx.in_pts, g_rows[:, :xyz_ch], ctx.in_scale)
                if ctx.needs_input_grad[1]:      # a ray's direction is encoded once for its spr samples: sum their rows first
                    g_dir = g_rows[:, xyz_ch:].reshape(-1, ctx.in_spr, dir_ch).sum(1)
                    g_vd = encoding_vjp(ctx.in_vd, g_dir, ctx.in_scale)
            ctx.in_params = ctx.in_pts = ctx.in_vd = None
        ctx.saved_acts = None
        return (g_pts, g_vd, g_embedded, g_cam) + (None,) * 3 + tuple(grads)


class SampleConstFn(torch.autograd.Function):
    """sample_pdf (run_nerf_helpers.py:241-284) -> plnerf_sample_const / plnerf_sample_const_bwd.  Differentiable
    with respect to `weights` (through the normalised cdf), which only the depth-supervised variant's
    piecewise-constant mode uses (depth_supervised_exps/model/run_nerf_helpers.py:343-394); on the NVS path the
    result is detached."""

    @staticmethod
    def forward(ctx, bins, weights, u):
        _expect(bins.dim() == 2, f"bins must be [rays, knots], got {tuple(bins.shape)}")
        R, B = bins.shape
        N = u.shape[-1]
        dev = bins.device
        _expect(tuple(weights.shape) == (R, B - 1), f"weights must be [{R}, {B - 1}], got {tuple(weights.shape)}")
        _expect(tuple(u.shape) in ((N,), (R, N)), f"u must be [{N}] or [{R}, {N}], got {tuple(u.shape)}")
        bins_c, w_c, u_c = _f32c(bins), _f32c(weights), _f32c(u)
        stride = N if u_c.dim() == 2 else 0
        out = torch.empty(R, N, device=dev)
        inds = torch.empty(R, N, device=dev, dtype=torch.int64)
        if R > 0:
            L.check(L.lib().plnerf_sample_const(
                L.dptr(bins_c, "bins"), L.dptr(w_c, "weights"), L.dptr(u_c, "u"), stride, R, B, N, L.dptr(out),
                L.dptr(inds, "inds", torch.int64), L.stream()), "plnerf_sample_const")
        ctx.save_for_backward(bins_c, w_c, u_c, inds)
        ctx.stride = stride
        ctx.mark_non_differentiable(inds)
        ctx.set_materialize_grads(False)
        return out, inds

    @staticmethod
    def backward(ctx, g_samples, g_inds):
        if g_samples is None:
            return None, None, None
        bins_c, w_c, u_c, inds = ctx.saved_tensors
        R, B = bins_c.shape
        N = inds.shape[-1]
        g_c = _f32c(g_samples)
        g_w = torch.empty(R, B - 1, device=bins_c.device)
        if R > 0:
            L.check(L.lib().plnerf_sample_const_bwd(
                L.dptr(bins_c), L.dptr(w_c), L.dptr(u_c), ctx.stride, L.dptr(inds, "inds", torch.int64), L.dptr(g_c),
                R, B, N, L.dptr(g_w), L.stream()), "plnerf_sample_const_bwd")
        g_bins = None
        if ctx.needs_input_grad[0]:
            # the bins carry a gradient (a ray batch that requires one): samples = b0 + t (b1 - b0) with t a function of the
            # weights and the draw only (run_nerf_helpers.py:266-282) -> (1 - t) to the lower bin, t to the upper one.  Off
            # every training path; a few torch launches (the two scatter-adds sum a bin's samples in no fixed order).
            # (t = (u - cdf) / (cdf step) cancels: evaluated in fp64 from the fp32 weights, then rounded once)
            w = w_c.double() + 1e-5
            cdf = torch.cumsum(w / torch.sum(w, -1, keepdim=True), -1)
            cdf = torch.cat([torch.zeros_like(cdf[..., :1]), cdf], -1)
            below, above = torch.clamp(inds - 1, min=0), torch.clamp(inds, max=B - 1)
            c0 = torch.gather(cdf, -1, below)
            denom = torch.gather(cdf, -1, above) - c0
            denom = torch.where(denom < 1e-5, torch.ones_like(denom), denom)
            t = (((u_c if u_c.dim() == 2 else u_c.expand(R, N)).double() - c0) / denom).float()
            g_bins = torch.zeros(R, B, device=bins_c.device).scatter_add_(-1, below, g_c * (1.0 - t)).scatter_add_(-1, above, g_c * t)
        return g_bins, g_w, None


def sample_const(bins, weights, u, want_inds=False):
    """plnerf_sample_const: bins [R,B], weights [R,B-1], u [R,N] or shared [N].  `samples` is differentiable with
    respect to `weights` and, when they ask, to `bins` (SampleConstFn)."""
    out, inds = SampleConstFn.apply(bins, weights, u)
    return (out, inds) if want_inds else out


class SamplePlFn(torch.autograd.Function):
    """sample_pdf_reformulation (run_nerf_helpers.py:364-445) -> plnerf_sample_pl / plnerf_sample_pl_bwd.
    Differentiable with respect to tau and T (what autograd derives for the reference: the interval
    search is piecewise constant in the weights) and, when they ask, to the bins z / near / far themselves
    (plnerf_sample_pl_bwd_rays: a ray batch that requires a gradient).  Outputs:
    samples, T_below, tau_below, bin_below, inds -- only `samples` carries a gradient."""

    @staticmethod
    def forward(ctx, z, weights, tau, T, near, far, u, zero_tol, eps):
        _expect(z.dim() == 2, f"z_vals must be [rays, samples], got {tuple(z.shape)}")
        R, S = z.shape
        N = u.shape[-1]
        dev = z.device
        _expect(tuple(weights.shape) == (R, S + 1), f"weights must be [{R}, {S + 1}], got {tuple(weights.shape)}")
        _expect(tuple(tau.shape) == (R, S + 2) and tuple(T.shape) == (R, S + 2), f"tau / T must be [{R}, {S + 2}]")
        _expect(near.numel() == R and far.numel() == R, f"near / far must hold one value per ray ({R})")
        _expect(tuple(u.shape) in ((N,), (R, N)), f"u must be [{N}] or [{R}, {N}], got {tuple(u.shape)}")
        z_c, w_c, tau_c, T_c = _f32c(z), _f32c(weights), _f32c(tau), _f32c(T)
        near_c, far_c, u_c = _f32c(near).reshape(-1), _f32c(far).reshape(-1), _f32c(u)
        stride = N if u_c.dim() == 2 else 0
        out = torch.empty(R, N, device=dev)
        Tb = torch.empty(R, N, device=dev)
        taub = torch.empty(R, N, device=dev)
        binb = torch.empty(R, N, device=dev)
        inds = torch.empty(R, N, device=dev, dtype=torch.int64)
        if R > 0:
            L.check(L.lib().plnerf_sample_pl(
                L.dptr(z_c, "z_vals"), L.dptr(w_c, "weights"), L.dptr(tau_c, "tau"), L.dptr(T_c, "T"),
                L.dptr(near_c, "near"), L.dptr(far_c, "far"), L.dptr(u_c, "u"), stride, R, S, N, float(zero_tol),
                float(eps), L.dptr(out), L.dptr(Tb), L.dptr(taub), L.dptr(binb),
                L.dptr(inds, "inds", torch.int64), L.stream()), "plnerf_sample_pl")
        ctx.save_for_backward(z_c, tau_c, T_c, near_c, far_c, u_c, inds)
        ctx.cfg = (stride, float(zero_tol), float(eps))
        ctx.near_shape, ctx.far_shape = near.shape, far.shape
        ctx.mark_non_differentiable(Tb, taub, binb, inds)
        ctx.set_materialize_grads(False)
        return out, Tb, taub, binb, inds

    @staticmethod
    def backward(ctx, g_samples, g_Tb, g_taub, g_binb, g_inds):
        if g_samples is None:
            return (None,) * 9
        z_c, tau_c, T_c, near_c, far_c, u_c, inds = ctx.saved_tensors
        stride, zero_tol, eps = ctx.cfg
        R, S = z_c.shape
        N = inds.shape[-1]
        g_c = _f32c(g_samples)
        g_tau = torch.empty(R, S + 2, device=z_c.device)
        g_T = torch.empty(R, S + 2, device=z_c.device)
        need = ctx.needs_input_grad
        if need[0] or need[4] or need[5]:
            # the bins carry a gradient (a ray batch that requires one): plnerf_sample_pl_bwd_rays also returns the knots' own
            g_k = torch.empty(R, S + 2, device=z_c.device)
            if R > 0:
                L.check(L.lib().plnerf_sample_pl_bwd_rays(
                    L.dptr(z_c), L.dptr(tau_c), L.dptr(T_c), L.dptr(near_c), L.dptr(far_c), L.dptr(u_c), stride,
                    L.dptr(inds, "inds", torch.int64), L.dptr(g_c), R, S, N, zero_tol, eps, L.dptr(g_tau), L.dptr(g_T),
                    L.dptr(g_k), L.stream()), "plnerf_sample_pl_bwd_rays")
            return (g_k[:, 1:-1] if need[0] else None, None, g_tau, g_T,
                    g_k[:, 0].reshape(ctx.near_shape) if need[4] else None,
                    g_k[:, -1].reshape(ctx.far_shape) if need[5] else None, None, None, None)
        if R > 0:
            L.check(L.lib().plnerf_sample_pl_bwd(
                L.dptr(z_c), L.dptr(tau_c), L.dptr(T_c), L.dptr(near_c), L.dptr(far_c), L.dptr(u_c), stride,
                L.dptr(inds, "inds", torch.int64), L.dptr(g_c), R, S, N, zero_tol, eps, L.dptr(g_tau), L.dptr(g_T),
                L.stream()), "plnerf_sample_pl_bwd")
        return None, None, g_tau, g_T, None, None, None, None, None


def sample_pl(z, weights, tau, T, near, far, u, zero_tol, eps, want_extras=False, want_inds=False):
    """plnerf_sample_pl; returns samples or (samples, T_below, tau_below, bin_below[, inds]).
    `samples` is differentiable with respect to tau and T (SamplePlFn)."""
    out, Tb, taub, binb, inds = SamplePlFn.apply(z, weights, tau, T, near, far, u, zero_tol, eps)
    if want_extras:
        return (out, Tb, taub, binb, inds) if want_inds else (out, Tb, taub, binb)
    return (out, inds) if want_inds else out


def merge_sort(z, z_new, near, far):
    """plnerf_merge_sort: sort(cat([z, clamp(z_new, near, far)]))."""
    R, S = z.shape
    N = z_new.shape[-1]
    out = torch.empty(R, S + N, device=z.device)
    if R == 0:
        return out
    # keep every (possibly copied) operand alive until the launch is enqueued: a temporary
    # freed mid-expression would hand its block to the next temporary
    z_c, zn_c = _f32c(z), _f32c(z_new)
    near_c, far_c = _f32c(near).reshape(-1), _f32c(far).reshape(-1)
    L.check(L.lib().plnerf_merge_sort(
        L.dptr(z_c, "z_vals"), L.dptr(zn_c, "z_samples"), L.dptr(near_c, "near"), L.dptr(far_c, "far"),
        R, S, N, L.dptr(out), L.stream()), "plnerf_merge_sort")
    return out


def stratified_z(near, far, t_vals, t_rand, lindisp=False):
    """plnerf_stratified_z: the coarse sample depths of run_plnerf.py:683-705 (no gradient: the ray batch carries
    none on this path).  near, far [R] or [R,1]; t_vals [S]; t_rand [R,S] or None."""
    near_c, far_c = _f32c(near).reshape(-1), _f32c(far).reshape(-1)
    t_c = _f32c(t_vals)
    r_c = _f32c(t_rand) if t_rand is not None else None
    R, S = near_c.shape[0], t_c.shape[0]
    out = torch.empty(R, S, device=near_c.device)
    L.check(L.lib().plnerf_stratified_z(
        L.dptr(near_c, "near"), L.dptr(far_c, "far"), L.dptr(t_c, "t_vals"), L.dptr(r_c, "t_rand"), R, S,
        int(bool(lindisp)), L.dptr(out), L.stream()), "plnerf_stratified_z")
    return out


def ray_points(rays_o, rays_d, z_vals):
    """plnerf_ray_points: pts = rays_o[..., None, :] + rays_d[..., None, :] * z_vals[..., :, None]
    (run_plnerf.py:708, :735), no gradient."""
    o_c, d_c, z_c = _f32c(rays_o), _f32c(rays_d), _f32c(z_vals)
    R, S = z_c.shape
    out = torch.empty(R, S, 3, device=z_c.device)
    L.check(L.lib().plnerf_ray_points(L.dptr(o_c, "rays_o"), L.dptr(d_c, "rays_d"), L.dptr(z_c, "z_vals"), R, S,
                                      L.dptr(out), L.stream()), "plnerf_ray_points")
    return out


_LINSPACE_CACHE = {}


def cpu_linspace(n, device):
    """torch.linspace(0, 1, n) as the reference's CPU path computes it (two-sided fma form),
    uploaded once per (n, device) -- so det=True draws are bit-identical to the oracle's."""
    key = (int(n), str(device))
    if key not in _LINSPACE_CACHE:
        _LINSPACE_CACHE[key] = torch.linspace(0.0, 1.0, steps=int(n), device="cpu").to(device)
    return _LINSPACE_CACHE[key]


def numpy_uniform(shape, device):
    """The reference's pytest=True draw: np.random.seed(0); np.random.rand(*shape) -> fp32."""
    np.random.seed(0)
    return torch.Tensor(np.random.rand(*shape)).to(device)


def embed_rows(pts, viewdirs, cam, n_freqs_xyz, n_freqs_dir, input_scale=1.0, bb_center=0.0, bb_scale=1.0):
    """plnerf_embed_rows: run_network's input assembly in one launch.  pts [R, S, 3]; viewdirs [R, 3] or None; cam
    [n_cam] or None (needs viewdirs).  Returns embedded [R * S, 3 + 6 fx (+ 3 + 6 fd + n_cam)]; no gradient (the
    positions carry none on this path; a trainable `cam` gets its gradient through MlpFn)."""
    R, S = pts.shape[0], pts.shape[1]
    pts_c = _f32c(pts).reshape(-1, 3)
    vd_c = None if viewdirs is None else _f32c(viewdirs)
    cam_c = None if (cam is None or cam.numel() == 0) else _f32c(cam).reshape(-1)
    n_cam = 0 if cam_c is None else cam_c.numel()
    C = 3 + 6 * n_freqs_xyz + (0 if vd_c is None else 3 + 6 * n_freqs_dir + n_cam)
    out = torch.empty(R * S, C, device=pts_c.device)
    if isinstance(bb_center, (tuple, list)) and len(bb_center) == 3:      # (host floats: no device read)
        center = (ctypes.c_float * 3)(*[float(v) for v in bb_center])
    else:
        c = torch.as_tensor(bb_center, dtype=torch.float32).reshape(-1).cpu()
        center = (ctypes.c_float * 3)(*[float(c[i if c.numel() == 3 else 0]) for i in range(3)])
    L.check(L.lib().plnerf_embed_rows(L.dptr(pts_c, "pts"), L.dptr(vd_c, "viewdirs"), L.dptr(cam_c, "cam"), R * S, S,
                                      int(n_freqs_xyz), int(n_freqs_dir), n_cam, float(input_scale), center,
                                      float(bb_scale), L.dptr(out), L.stream()), "plnerf_embed_rows")
    return out


def joint_choice(pred_hyp, target_h, mask, threshold=0.0, group=None):
    """The is_joint hypothesis per point column for a batch sharded over `group`: [n_points] int32.  Column sums of this
    shard (plnerf_depth_joint_sums), ONE all-reduce (SUM, fp64 [n_hyp, n_points]), then the reference's comparison: fp32
    means over the global rays, first minimum on ties (torch.min).  Identical on every rank."""
    hyp_c, th_c = _f32c(pred_hyp), _f32c(target_h)
    R, P, H, PT = hyp_c.shape[0], hyp_c.shape[1], th_c.shape[0], th_c.shape[-1]
    sums = torch.empty(H, P, device=hyp_c.device, dtype=torch.float64)
    L.check(L.lib().plnerf_depth_joint_sums(L.dptr(hyp_c, "pred_hyp"), L.dptr(th_c, "target_h"), L.dptr(mask, "mask"), R, P, H,
                                            PT, float(threshold), L.dptr(sums, "sums", torch.float64), L.stream()),
            "plnerf_depth_joint_sums")
    rays = torch.tensor([float(R)], device=hyp_c.device, dtype=torch.float64)
    if torch.distributed.is_initialized() and torch.distributed.get_world_size(group) > 1:
        both = torch.cat([sums.reshape(-1), rays])
        torch.distributed.all_reduce(both, op=torch.distributed.ReduceOp.SUM, group=group)
        sums, rays = both[:-1].reshape(H, P), both[-1:]
    return (sums / rays).float().argmin(0).to(torch.int32).contiguous()


def depth_loss_and_grads(rgb, rgb0, target, pred_hyp, target_h, weight, threshold=0.0, mask=None, is_joint=False,
                         group=None, sharded=False):
    """plnerf_depth_loss: the depth-supervised loop's loss and its three gradients in one launch (is_joint: the
    hypothesis is chosen per image -- per point column -- instead of per ray, model/run_nerf_helpers.py:72-77).
    Returns (loss5 = [total, img, img0, space carving, psnr], g_rgb, g_rgb0, g_hyp); rgb0 / pred_hyp may be None.

    sharded (is_joint only): these rays are one rank's shard of a global batch.  The reference chooses the hypothesis
    from the mean over ALL rays of the batch (nn.DataParallel gathers the outputs before the loss,
    run_nerf_sample_based_depth.py:564, 585), so the shards' column sums (plnerf_depth_joint_sums, fp64 [n_hyp, n_points],
    a few KB) are added over `group` before the minimum is taken -- every rank then differentiates the same hypotheses,
    and the ranks' averaged loss is the one-rank loss of the global batch."""
    rgb_c, t_c = _f32c(rgb), _f32c(target)
    rgb0_c = None if rgb0 is None else _f32c(rgb0)
    _expect(rgb_c.shape == t_c.shape and rgb_c.dim() == 2 and rgb_c.shape[1] == 3, "rgb / target must be [R, 3]")
    R = rgb_c.shape[0]
    hyp_c = th_c = mask_c = g_h = None
    P = H = PT = 1
    if pred_hyp is not None:
        hyp_c, th_c = _f32c(pred_hyp), _f32c(target_h)
        P, H, PT = hyp_c.shape[1], th_c.shape[0], th_c.shape[-1]
        _expect(hyp_c.shape[0] == R and th_c.dim() == 3 and th_c.shape[1] == R and PT in (1, P),
                f"pred_hyp [R, P] / target_h [H, R, 1 or P]: got {tuple(hyp_c.shape)} / {tuple(th_c.shape)}")
        g_h = torch.empty_like(hyp_c)
        mask_c = None if mask is None else _f32c(mask).reshape(-1)
        _expect(mask_c is None or mask_c.numel() == R, "mask must hold one value per ray")
    loss5 = torch.empty(5, device=rgb_c.device)
    g1 = torch.empty_like(rgb_c)
    g0 = None if rgb0_c is None else torch.empty_like(rgb_c)
    ws = _loss_workspace(rgb_c.device, L.DEPTH_LOSS_WORKSPACE_BYTES, "depth")
    choice = None
    if is_joint and sharded and pred_hyp is not None:
        choice = joint_choice(hyp_c, th_c, mask_c, threshold, group)
    L.check(L.lib().plnerf_depth_loss(L.dptr(rgb_c, "rgb"), L.dptr(rgb0_c, "rgb0"), L.dptr(t_c, "target"),
                                      L.dptr(hyp_c, "pred_hyp"), L.dptr(th_c, "target_h"), L.dptr(mask_c, "mask"), R, P, H,
                                      PT, int(bool(is_joint)), L.dptr(choice, "joint_choice", torch.int32), float(weight),
                                      float(threshold), L.dptr(loss5), L.dptr(g1),
                                      L.dptr(g0),
                                      L.dptr(g_h), L.dptr(ws, "workspace", torch.float64), L.stream()), "plnerf_depth_loss")
    return loss5, g1, g0, g_h
