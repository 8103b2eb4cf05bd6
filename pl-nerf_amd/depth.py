"""Depth-supervised variant of the path (BASELINE config 5): the caller-side
mirror of depth_supervised_exps/run_nerf_sample_based_depth.py and
depth_supervised_exps/model/run_nerf_helpers.py, on the same HIP kernels.

What differs from the NVS path (render.py):
  * the encoder multiplies by pi (model/run_nerf_helpers.py:123) and the default widths are
    multires 9 / multires_views 0 -> input_ch 57, input_ch_views 3 (run_nerf_sample_based_depth.py:
    1306-1309): the encoding is evaluated here and handed to the fused MLP as `embedded`;
  * positions pass through (x - bb_center) * bb_scale first (:56);
  * the density channel goes through softplus(beta=10) (model/run_nerf_helpers.py:200);
  * render_rays draws a third set of samples, `pred_hyp`, from the final weights WITHOUT detaching them
    (:923-934): the space-carving loss back-propagates through the sampler (plnerf_sample_pl_bwd) and
    the transmittance / density knots (g_tau, g_T of plnerf_quad_bwd);
  * one Adam over both networks, gradient values clipped to 0.1 (:1155-1157).
"""
import os
import sys

import numpy as np
import torch

from . import _lib as L
from . import functional as Fn
from . import raybatch as RB
from .nerf import Embedder, NeRF
from .optim import FlatAdam
from .render import (MAX_ROWS_PER_LAUNCH, _draw_noise, _refuse_unsupported, _rgb_sigma, batchify, raw2outputs as _raw2outputs, sample_pdf,
                     sample_pdf_reformulation)

_RENDER = sys.modules[__name__.rsplit(".", 1)[0] + ".render"]      # (the package attribute `render` is the function)


def _nvs_draw_u(*args):
    """The importance draw of the NVS sampler (render._draw_u), looked up at call time like sample_pdf_reformulation
    itself does (a test that injects its draws there reaches both routes)."""
    return _RENDER._draw_u(*args)


def get_embedder(multires, i=0):
    """model/run_nerf_helpers.py:132-148: gamma(x) with sin/cos(x * pi * 2^k)."""
    if i == -1:
        return torch.nn.Identity(), 3
    emb = Embedder(include_input=True, input_dims=3, max_freq_log2=multires - 1, num_freqs=multires,
                   log_sampling=True, periodic_fns=[torch.sin, torch.cos], input_scale=np.pi)
    return emb, emb.out_dim


def _kernel_encoding(embed_fn, embeddirs_fn, viewdirs):
    """(fx, fd, input scale) when plnerf_embed_rows evaluates these two encoders, else None: gamma(x) = [x, sin / cos of
    x s 2^k, k < L] with s = 1 (run_nerf_helpers.py:24-54) or pi (depth_supervised_exps/model/run_nerf_helpers.py:123)."""
    def freqs(e):
        if not isinstance(e, Embedder):
            return None
        sc = 1.0 if e.input_scale is None else float(e.input_scale)
        ok = (e.input_dims == 3 and e.include_input and e.log_sampling and list(e.periodic_fns) == [torch.sin, torch.cos]
              and e.max_freq_log2 == e.num_freqs - 1 and 0 <= e.num_freqs <= 16)
        return (e.num_freqs, sc) if ok else None
    fx = freqs(embed_fn)
    if fx is None:
        return None
    if viewdirs is None:
        return fx[0], 0, fx[1]
    fd = freqs(embeddirs_fn)
    if fd is None:
        return None
    scales = {sc for n, sc in (fx, fd) if n > 0}      # (an encoder without frequency bands has no use for its scale)
    if len(scales) > 1:
        return None
    return fx[0], fd[0], (scales.pop() if scales else 1.0)


FUSE_STAGES = True      # (tests switch it off to compare the one-launch stages with the separate launches, bit for bit)
# Tests only (tests/test_gpu_fullsize.py), as render.STAGE_TAP: a dict that render_rays fills with its intermediate tensors
# (coarse weights / tau / T and importance samples, final weights / tau / T and the hypotheses' search indices); while it
# is set the stages run as their separate launches, which is where those tensors exist in HBM.
STAGE_TAP = None
_BOX_CACHE = {}


def _host_box(bb_center, bb_scale):
    """The bounding-box affine as host floats ((cx, cy, cz), scale, is-identity).  The reference script keeps the box as
    device tensors (run_nerf_sample_based_depth.py:52-56); reading them costs a host synchronisation, which must not
    happen on every network evaluation of a training step: resolved once per (tensor, version)."""
    def key(x):
        return (id(x), x._version) if isinstance(x, torch.Tensor) else ("v", repr(x))
    k = (key(bb_center), key(bb_scale))
    hit = _BOX_CACHE.get(k)
    hit = hit[0] if hit is not None else None
    if hit is None:
        c = torch.as_tensor(bb_center, dtype=torch.float32).reshape(-1).cpu()
        center = tuple(float(c[i if c.numel() == 3 else 0]) for i in range(3))
        scale = float(torch.as_tensor(bb_scale))
        hit = (center, scale, scale == 1.0 and not any(center))
        if len(_BOX_CACHE) > 64:
            _BOX_CACHE.clear()
        # (the entry keeps the two objects alive: an id() can otherwise be handed to a new tensor after the old one died)
        _BOX_CACHE[k] = (hit, bb_center, bb_scale)
    return hit


def run_network(inputs, viewdirs, embedded_cam, fn, embed_fn, embeddirs_fn, bb_center, bb_scale, netchunk=1024 * 64):
    """run_nerf_sample_based_depth.py:52-68: bounding-box affine, encodings (positions | per-ray direction | per-image
    camera code, each repeated over the ray's samples), then the MLP in row chunks.  inputs [R, S, 3].

    On the GPU the whole input assembly is ONE launch (plnerf_embed_rows) instead of ~100 element-wise / cat launches
    per network evaluation (1.2 ms of a 10.8 ms depth-supervised step, profiles/r03_depth_kernel_stats_before.csv).  A
    camera code that requires grad (trained through the network input, :1091-1093, 1122-1123; optimised alone,
    :311-345) is handed to the MLP as `cam` and gets its gradient there (functional.MlpFn)."""
    R, S = inputs.shape[0], inputs.shape[1]
    on_hip = isinstance(fn, NeRF) and fn.is_supported() and inputs.is_cuda
    # `netchunk` "does not affect final results" (rows are independent): on the HIP MLP the rows go in launches as large
    # as the saved-activation buffer allows instead of the reference's 65,536-row chunks (a training step would
    # otherwise pay one forward, one backward and one weight-gradient reduction PER CHUNK)
    if on_hip:
        netchunk = max(int(netchunk), MAX_ROWS_PER_LAUNCH)
    enc = _kernel_encoding(embed_fn, embeddirs_fn, viewdirs) if on_hip else None
    cam = None if (viewdirs is None or embedded_cam is None or embedded_cam.numel() == 0) else embedded_cam
    if enc is not None and not (torch.is_grad_enabled() and (inputs.requires_grad or
                                                             (viewdirs is not None and viewdirs.requires_grad))):
        fx, fd, scale = enc
        bb_center, bb_scale, identity_box = _host_box(bb_center, bb_scale)      # (host floats: no sync per evaluation)
        if (cam is None and identity_box and fn.has_fused_encoding() and fn.input_ch == 3 + 6 * fx and
                (not fn.use_viewdirs or fn.input_ch_views == 3 + 6 * fd)):
            # the encoding in the MLP kernel's own prologue, as on the NVS path: no `embedded` matrix at all
            rays_per_launch = max(1, MAX_ROWS_PER_LAUNCH // max(S, 1))
            if not torch.is_grad_enabled():      # (inference saves nothing: one launch whatever the size)
                rays_per_launch = max(R, 1)
            else:                                # equal shares (a remainder launch of a few rows costs a whole tile walk)
                rays_per_launch = -(-R // max(-(-R // rays_per_launch), 1)) if R > 0 else 1
            vd = viewdirs if fn.use_viewdirs else None
            outs = [fn.query(inputs[i:i + rays_per_launch], None if vd is None else vd[i:i + rays_per_launch],
                             input_scale=scale) for i in range(0, R, rays_per_launch)]
            return outs[0] if len(outs) == 1 else torch.cat(outs, 0)
        embedded = Fn.embed_rows(inputs, viewdirs, cam, fx, fd, input_scale=scale, bb_center=bb_center,
                                 bb_scale=bb_scale)
        cam_grad = cam if (cam is not None and torch.is_grad_enabled() and cam.requires_grad) else None
        blocks = [fn(block, cam=cam_grad) if cam_grad is not None else fn(block)
                  for block in torch.split(embedded, netchunk, dim=0)]
        raw = blocks[0] if len(blocks) == 1 else torch.cat(blocks, 0)
        return raw.reshape(*inputs.shape[:-1], raw.shape[-1])
    columns = [embed_fn(((inputs.reshape(-1, inputs.shape[-1]) - bb_center) * bb_scale))]
    if viewdirs is not None:
        per_ray = embeddirs_fn(viewdirs)                              # row-wise encoder: encode once per ray ...
        columns.append(per_ray[:, None, :].expand(R, S, per_ray.shape[-1]).reshape(R * S, -1))   # ... then repeat
        columns.append(embedded_cam.reshape(1, -1).expand(R * S, embedded_cam.shape[0]))
    raw = batchify(fn, netchunk)(torch.cat(columns, -1))
    return raw.reshape(*inputs.shape[:-1], raw.shape[-1])


def raw2outputs(raw, z_vals, near, far, rays_d, mode, color_mode, raw_noise_std=0, pytest=False, white_bkgd=False,
                farcolorfix=False):
    """run_nerf_sample_based_depth.py:709-773.  `farcolorfix` is accepted and ignored there."""
    return _raw2outputs(raw, z_vals, near, far, rays_d, mode, color_mode, raw_noise_std, pytest=pytest,
                        white_bkgd=white_bkgd, farcolorfix=False)


def _draw_u(n_rays, N_samples, det, pytest, load_u, joint, device):
    """The draw of the *_return_u samplers (model/run_nerf_helpers.py:619-638; joint: 792-812)."""
    if load_u is not None:
        return load_u
    draws = None if pytest else Fn.DRAWS
    if det:
        u = Fn.cpu_linspace(N_samples, device).expand(n_rays, N_samples)
    elif joint:      # one row for the whole image
        row = (_joint_row(draws, N_samples, device) if draws is not None else torch.rand(N_samples, device=device))
        u = row.unsqueeze(0).repeat(n_rays, 1)
    elif draws is not None:
        u = draws.uniform(n_rays, N_samples, Fn.FineEpilogueFn.HYP_STREAM, device)
    else:
        u = torch.rand(n_rays, N_samples, device=device)
    if pytest:
        np.random.seed(0)
        if det:
            u = torch.Tensor(np.broadcast_to(np.linspace(0., 1., N_samples), [n_rays, N_samples]).copy()).to(device)
        else:
            u = torch.Tensor(np.random.rand(n_rays, N_samples)).to(device)
    return u


def _joint_row(draws, n, device):
    """The is_joint draw -- ONE row of u for the whole image -- from the counters of global ray 0, so that every rank of
    a sharded batch sees the same row."""
    keep = draws.ray_id0, draws.chunk_offset
    draws.ray_id0, draws.chunk_offset = 0, 0
    try:
        return draws.uniform(1, n, Fn.FineEpilogueFn.HYP_STREAM, device).reshape(-1)
    finally:
        draws.ray_id0, draws.chunk_offset = keep


def sample_pdf_reformulation_return_u(bins, weights, tau, T, near, far, N_samples, det=False, pytest=False,
                                      load_u=None, quad_solution_v2=True, zero_threshold=1e-4, epsilon_=1e-3,
                                      joint=False):
    """model/run_nerf_helpers.py:607-692 (and :780- for joint=True).  Returns (samples, T_below, tau_below,
    bin_below, u); `samples` is differentiable with respect to tau and T."""
    u = _draw_u(bins.shape[0], N_samples, det, pytest, load_u, joint, bins.device).contiguous()
    s, Tb, taub, binb = Fn.sample_pl(bins, weights, tau, T, near, far, u, zero_threshold, epsilon_, want_extras=True)
    return s, Tb, taub, binb, u


def sample_pdf_return_u(bins, weights, N_samples, det=False, pytest=False, load_u=None, joint=False):
    """model/run_nerf_helpers.py:343-394 (piecewise-constant mode; :446- for joint=True).  Returns (samples, u);
    `samples` is differentiable with respect to `weights` (plnerf_sample_const_bwd)."""
    u = _draw_u(bins.shape[0], N_samples, det, pytest, load_u, joint, bins.device).contiguous()
    return Fn.sample_const(bins, weights, u), u


def _draw_t_rand(n_rays, n_samples, pytest, device):
    """The stratified jitter of run_nerf_sample_based_depth.py:781-788: np.random.seed(0) draws under pytest; else from
    an installed functional.DrawSource (counter-based on the global ray id: the step does not depend on how the batch
    is sharded), else torch.rand as the reference."""
    if pytest:
        return Fn.numpy_uniform([n_rays, n_samples], device)
    if Fn.DRAWS is not None:
        return Fn.DRAWS.uniform(n_rays, n_samples, Fn.DrawSource.T_RAND, device)
    return torch.rand(n_rays, n_samples, device=device)


def perturb_z_vals(z_vals, pytest):
    """run_nerf_sample_based_depth.py:775-790."""
    mids = .5 * (z_vals[..., 1:] + z_vals[..., :-1])
    upper = torch.cat([mids, z_vals[..., -1:]], -1)
    lower = torch.cat([z_vals[..., :1], mids], -1)
    t_rand = _draw_t_rand(z_vals.shape[0], z_vals.shape[1], pytest, z_vals.device)
    return lower + (upper - lower) * t_rand


def render_rays(ray_batch, use_viewdirs, network_fn, network_query_fn, N_samples, mode, color_mode,
                precomputed_z_samples=None, embedded_cam=None, retraw=False, lindisp=False, perturb=0.,
                N_importance=0, network_fine=None, raw_noise_std=0., verbose=False, pytest=False, white_bkgd=False,
                is_joint=False, cached_u=None, scale_sample_gradient=False, quad_solution_v2=False, zero_tol=1e-4,
                epsilon=1e-3, farcolorfix=False):
    """run_nerf_sample_based_depth.py:792-958.  Returns the reference's dict: rgb_map, disp_map, acc_map,
    depth_map, z_vals, weights, pred_hyp, u (+ raw; + rgb0, disp0, acc0, depth0, z_vals0, weights0, z_std)."""
    dev = ray_batch.device
    N_rays = ray_batch.shape[0]
    if isinstance(ray_batch, RB.RayColumns):      # (what plnerf_select_rays writes: no packing, no slice copies)
        rays_o, rays_d, near, far = ray_batch.rays_o, ray_batch.rays_d, ray_batch.near.reshape(-1, 1), \
            ray_batch.far.reshape(-1, 1)
        viewdirs = ray_batch.viewdirs if use_viewdirs else None
    else:
        rays_o, rays_d = ray_batch[:, 0:3].contiguous(), ray_batch[:, 3:6].contiguous()
        viewdirs = ray_batch[:, 8:11].contiguous() if use_viewdirs else None
        near, far = ray_batch[:, 6:7].contiguous(), ray_batch[:, 7:8].contiguous()
    # A ray batch that requires a gradient (no training path of the reference has one): as in render.py, the glue then runs
    # as the reference's own torch expressions, the quadrature and the samplers return the geometry's gradient
    # (plnerf_quad_bwd_rays; plnerf_sample_pl_bwd_rays / SampleConstFn's bins -- here the depth hypotheses STAY attached to the
    # sampler's bins, :923-934), the networks their inputs' (plnerf_mlp_input_grad).
    rays_grad = torch.is_grad_enabled() and any(
        t is not None and t.requires_grad for t in (rays_o, rays_d, near, far, viewdirs))
    t_vals = Fn.cpu_linspace(N_samples, dev)
    fused_glue = ray_batch.is_cuda and N_rays > 0 and not rays_grad
    draws = None if pytest else Fn.DRAWS      # counter-based draws inside the consuming kernels (functional.DrawSource)
    if draws is not None:
        draws.noise_calls = 0
    # The stages between and behind the two network evaluations as ONE launch each, like the NVS path's (piecewise-linear
    # mode): plnerf_coarse_epilogue (raw2outputs + importance sampling + clamp + sort + positions) and
    # plnerf_fine_epilogue (raw2outputs + the hypotheses' sampler + z_std); bit-identical to the separate calls below.
    tap = STAGE_TAP
    fused = FUSE_STAGES and fused_glue and mode == "linear" and color_mode in ("midpoint", "left") and tap is None
    if fused_glue:
        # depths, jitter and positions in one launch (plnerf_coarse_samples: bit-identical to the expressions below,
        # which are :775-790 and run_plnerf.py:683-708 alike); the jitter from the reference's draw, or in the kernel
        t_rand = _draw_t_rand(N_rays, N_samples, pytest, dev) if (perturb > 0. and draws is None) else None
        z_vals, pts = Fn.coarse_samples(rays_o, rays_d, near, far, t_vals, t_rand, lindisp, perturb > 0., draws)
    else:
        if not lindisp:
            z_vals = near * (1. - t_vals) + far * t_vals
        else:
            z_vals = 1. / (1. / near * (1. - t_vals) + 1. / far * t_vals)
        if perturb > 0.:
            z_vals = perturb_z_vals(z_vals, pytest)
        pts = rays_o[..., None, :] + rays_d[..., None, :] * z_vals[..., :, None]
    raw = network_query_fn(pts, viewdirs, embedded_cam, network_fn)
    if tap is not None:
        tap.update(z_vals0=z_vals, raw0=raw)

    def last_stage(raw, z_vals, n, load_u):
        """raw2outputs of the final pass + the depth hypotheses on its weights (:909-934), one launch."""
        det = perturb == 0.
        if load_u is not None or det or pytest or draws is None:
            u = _draw_u(N_rays, n, det, pytest, load_u, is_joint, dev).contiguous()
        else:      # drawn inside the kernel; is_joint: one row from the counters of ray 0
            u = _joint_row(draws, n, dev) if is_joint else None
        rgb, disp, acc, depth, w, tau, T, hyp, u_used, _, z_std = Fn.FineEpilogueFn.apply(
            _rgb_sigma(raw), z_vals, near, far, rays_d, _draw_noise(raw, raw_noise_std, pytest), u, n, color_mode,
            white_bkgd, False, zero_tol, epsilon, draws)
        return rgb, disp, acc, w, depth, hyp, u_used, z_std

    if fused and N_importance == 0:
        rgb_map, disp_map, acc_map, weights, depth_map, pred_depth_hyp, u, _ = last_stage(raw, z_vals, N_samples, None)
    elif fused:
        z_vals_0 = z_vals
        det = perturb == 0.
        u0 = _nvs_draw_u([N_rays], N_importance, det, pytest, dev) if (pytest or det or draws is None) else None
        rgb_map_0, disp_map_0, acc_map_0, depth_map_0, z_vals, pts, _, weights_0 = Fn.CoarseEpilogueFn.apply(
            _rgb_sigma(raw), z_vals, near, far, rays_o, rays_d, _draw_noise(raw, raw_noise_std, pytest), u0,
            N_importance, color_mode, white_bkgd, False, zero_tol, epsilon, draws, True)
        run_fn = network_fn if network_fine is None else network_fine
        raw = network_query_fn(pts, viewdirs, embedded_cam, run_fn)
        rgb_map, disp_map, acc_map, weights, depth_map, pred_depth_hyp, u, z_std = last_stage(raw, z_vals, N_importance,
                                                                                             cached_u)
    else:
        rgb_map, disp_map, acc_map, weights, depth_map, tau, T = raw2outputs(
            raw, z_vals, near, far, rays_d, mode, color_mode, raw_noise_std, pytest=pytest, white_bkgd=white_bkgd,
            farcolorfix=farcolorfix)

    def hypotheses(z_vals, weights, tau, T, n, load_u):
        if mode == "linear" and tap is not None:
            u = _draw_u(N_rays, n, perturb == 0., pytest, load_u, is_joint, dev).contiguous()
            s, inds = Fn.sample_pl(z_vals, weights, tau, T, near, far, u, zero_tol, epsilon, want_inds=True)
            tap.update(weights_full=weights, tau=tau, T=T, hyp_inds=inds)
        elif mode == "linear":
            s, _, _, _, u = sample_pdf_reformulation_return_u(
                z_vals, weights, tau, T, near, far, n, det=(perturb == 0.), pytest=pytest, load_u=load_u,
                quad_solution_v2=quad_solution_v2, zero_threshold=zero_tol, epsilon_=epsilon, joint=is_joint)
        elif mode == "constant":
            z_mid = .5 * (z_vals[..., 1:] + z_vals[..., :-1])
            s, u = sample_pdf_return_u(z_mid, weights[..., 1:-1], n, det=(perturb == 0.), pytest=pytest,
                                       load_u=load_u, joint=is_joint)
        else:
            raise ValueError("mode must be 'linear' or 'constant'")
        return s, u

    if fused:
        pass
    elif N_importance == 0:
        pred_depth_hyp, u = hypotheses(z_vals, weights, tau, T, N_samples, None)
    else:
        rgb_map_0, disp_map_0, acc_map_0, depth_map_0, z_vals_0, weights_0 = \
            rgb_map, disp_map, acc_map, depth_map, z_vals, weights
        if mode == "linear" and tap is not None:
            u0 = _nvs_draw_u(z_vals.shape[:-1], N_importance, perturb == 0., pytest, dev)
            z_samples, inds0 = Fn.sample_pl(z_vals, weights, tau, T, near, far, u0, zero_tol, epsilon, want_inds=True)
            tap.update(weights0_full=weights, tau0=tau, T0=T, u0=u0, inds0=inds0, z_samples=z_samples)
        elif mode == "linear":
            with torch.set_grad_enabled(torch.is_grad_enabled() and not rays_grad):      # (detached below either way)
                z_samples = sample_pdf_reformulation(z_vals, weights, tau, T, near, far, N_importance, det=(perturb == 0.),
                                                     pytest=pytest, quad_solution_v2=quad_solution_v2,
                                                     zero_threshold=zero_tol, epsilon_=epsilon)[0]
        else:
            z_mid = .5 * (z_vals[..., 1:] + z_vals[..., :-1])
            z_samples = sample_pdf(z_mid, weights[..., 1:-1], N_importance, det=(perturb == 0.), pytest=pytest)
        z_samples = z_samples.detach()
        if rays_grad:      # :902-906 as written: the clamp's bounds and the coarse depths carry the batch's gradient
            z_vals, order = torch.sort(torch.cat([z_vals, torch.clamp(z_samples, near, far)], -1), -1)
            if tap is not None:
                tap.update(sort_order=order)      # (ties: render.py)
        else:
            z_vals = Fn.merge_sort(z_vals, z_samples, near, far)          # clamp + cat + sort (:902-906)
        if fused_glue:
            pts = Fn.ray_points(rays_o, rays_d, z_vals)
        else:
            pts = rays_o[..., None, :] + rays_d[..., None, :] * z_vals[..., :, None]
        run_fn = network_fn if network_fine is None else network_fine
        raw = network_query_fn(pts, viewdirs, embedded_cam, run_fn)
        rgb_map, disp_map, acc_map, weights, depth_map, tau, T = raw2outputs(
            raw, z_vals, near, far, rays_d, mode, color_mode, raw_noise_std, pytest=pytest, white_bkgd=white_bkgd)
        pred_depth_hyp, u = hypotheses(z_vals, weights, tau, T, N_importance, cached_u)
        z_std = torch.std(pred_depth_hyp, dim=-1, unbiased=False)

    if mode == "linear":
        weights = weights[..., 1:]
    ret = {'rgb_map': rgb_map, 'disp_map': disp_map, 'acc_map': acc_map, 'depth_map': depth_map,
           'z_vals': z_vals, 'weights': weights, 'pred_hyp': pred_depth_hyp, 'u': u}
    if retraw:
        ret['raw'] = raw
    if N_importance > 0:
        ret['rgb0'] = rgb_map_0
        ret['disp0'] = disp_map_0
        ret['acc0'] = acc_map_0
        ret['depth0'] = depth_map_0
        ret['z_vals0'] = z_vals_0
        ret['weights0'] = weights_0
        ret['z_std'] = z_std
    return ret


def get_ray_dirs(H, W, intrinsic, c2w, coords=None):
    """model/run_nerf_helpers.py:243-257: intrinsic = (fx, fy, cx, cy); pixel CENTRES (+0.5) and a flipped image
    row, unlike the NVS script's get_rays.  coords [n, 2] = (row, col) selects pixels."""
    fx, fy, cx, cy = intrinsic[0], intrinsic[1], intrinsic[2], intrinsic[3]
    if coords is None:
        dev = c2w.device
        cols, rows = torch.meshgrid(torch.linspace(0, W - 1, W, device=dev), torch.linspace(0, H - 1, H, device=dev),
                                    indexing='ij')
        cols, rows = cols.t(), rows.t()
    else:
        cols, rows = coords[:, 1], coords[:, 0]
    dirs = torch.stack([((cols + 0.5) - cx) / fx, (H - (rows + 0.5) - cy) / fy, -torch.ones_like(cols)], -1)
    return torch.sum(dirs[..., None, :] * c2w[:3, :3], -1)


def get_rays(H, W, intrinsic, c2w, coords=None):
    """model/run_nerf_helpers.py:259-263."""
    rays_d = get_ray_dirs(H, W, intrinsic, c2w, coords)
    return c2w[:3, -1].expand(rays_d.shape), rays_d


def batchify_rays(rays_flat, chunk=1024 * 32, use_viewdirs=False, **kwargs):
    """run_nerf_sample_based_depth.py:71-83."""
    return RB.map_row_chunks(lambda rows, first_row: render_rays(rows, use_viewdirs, **kwargs), rays_flat, chunk)


def render(H, W, intrinsic, chunk=1024 * 32, rays=None, c2w=None, ndc=True, near=0., far=1., with_5_9=False,
           use_viewdirs=False, c2w_staticcam=None, rays_depth=None, **kwargs):
    """run_nerf_sample_based_depth.py:85-160 (render_hyp, :162-248, is the same function).  Returns
    [rgb_map, disp_map, acc_map, extras]; `ndc` is accepted and unused, as there.  `rays` = (origins, directions) or
    (origins, directions, depth columns); a full view comes from `c2w`, optionally centre-cropped to 5.33:9."""
    if c2w is not None:
        rays_o, rays_d = get_rays(H, W, intrinsic, c2w)
        if with_5_9:
            crop = int(H / 9. * 16. / 3.)
            crop -= crop % 2
            first = (W - crop) // 2
            rays_o, rays_d, W = rays_o[:, first:first + crop, :], rays_d[:, first:first + crop, :], crop
    elif rays.shape[0] == 2:
        rays_o, rays_d = rays
    else:
        rays_o, rays_d, rays_depth = rays
    extra = []
    if use_viewdirs:
        extra.append(RB.unit_directions(rays_d))
        if c2w_staticcam is not None:
            rays_o, rays_d = get_rays(H, W, intrinsic, c2w_staticcam)
    if rays_depth is not None:
        extra.append(rays_depth.reshape(-1, 3).float())
    rows, lead = RB.pack_rays(rays_o, rays_d, near, far, extra)
    return RB.unflatten_outputs(batchify_rays(rows, chunk, use_viewdirs, **kwargs), lead)


render_hyp = render


class _SpaceCarvingFn(torch.autograd.Function):
    """compute_space_carving_loss on the GPU: plnerf_depth_loss's carving term and its gradient (the image terms of that
    launch run on a dummy pixel and are discarded)."""

    @staticmethod
    def forward(ctx, pred, target_h, mask, threshold, is_joint):
        # (a dummy image pair one apart: the discarded image term's psnr stays finite)
        dummy, other = pred.new_zeros(pred.shape[0], 3), pred.new_ones(pred.shape[0], 3)
        # weight 1: loss5[3] is the unweighted term, g_hyp its gradient
        loss5, _, _, g_hyp = Fn.depth_loss_and_grads(dummy, None, other, pred, target_h, 1.0, threshold=threshold, mask=mask,
                                                     is_joint=is_joint)
        ctx.g = g_hyp
        return loss5[3]

    @staticmethod
    def backward(ctx, g):
        return ctx.g * g, None, None, None, None


def compute_space_carving_loss(pred_depth, target_hypothesis, is_joint=False, mask=None, norm_p=2, threshold=0.0):
    """model/run_nerf_helpers.py:52-86.  pred_depth [n_rays, n_points]; target_hypothesis [n_hyp, n_rays, 1 or
    n_points]; returns the scalar loss, differentiable with respect to pred_depth.  The distance is a p-norm over a
    trailing axis of length ONE, i.e. |pred - target| for every p >= 1; the reductions (per ray: min over hypotheses,
    mean; is_joint: mean over rays, min over hypotheses per point, mean) run in plnerf_depth_loss."""
    if not norm_p >= 1:
        raise ValueError("norm_p must be >= 1 (the norm runs over an axis of length one: it is the absolute value)")
    if not pred_depth.is_cuda:
        raise RuntimeError(f"plnerf_amd: compute_space_carving_loss runs in plnerf_depth_loss and needs GPU tensors, got "
                           f"{pred_depth.device} (there is no CPU fallback)")
    return _SpaceCarvingFn.apply(pred_depth, target_hypothesis, mask, float(threshold), bool(is_joint))


def get_space_carving_idx(pred_depth, target_hypothesis, is_joint=False, mask=None, norm_p=2, threshold=0.0):
    """model/run_nerf_helpers.py:19-49: which hypothesis compute_space_carving_loss would pick, for the caller's
    hypothesis cache -- per ray and point [H, W, n_points], or (is_joint) the one index of the whole image repeated to
    [H, W, 1].  pred_depth [H, W, n_points]; target_hypothesis [n_hyp, H, W, 1].  (A data-pipeline helper, off the
    training step: a few torch reductions.)"""
    d = (pred_depth[None] - target_hypothesis).abs()                     # [n_hyp, H, W, n_points]
    if mask is not None:                                                  # one value per pixel, [H, W] or flat
        d = d * mask.reshape(1, pred_depth.shape[0], pred_depth.shape[1], 1)
    if threshold > 0:
        d = d.masked_fill(d < threshold, 0.0)
    if is_joint:
        best = d.flatten(1).mean(1).argmin()
        return best.expand(pred_depth.shape[0], pred_depth.shape[1], 1)
    return d.argmin(0)


def create_nerf(args, scene_render_params=None, device=None):
    """run_nerf_sample_based_depth.py:547-644: (render_kwargs_train, render_kwargs_test, start, grad_vars,
    optimizer).  One Adam over the parameters of both networks; no nn.DataParallel (ray shards go through
    dp.py instead, one process per GPU)."""
    device = RB.default_device(device)
    embed_fn, input_ch = get_embedder(args.multires, args.i_embed)
    embeddirs_fn, input_ch_views = (get_embedder(args.multires_views, args.i_embed) if args.use_viewdirs
                                    else (None, 0))

    def network(depth, width):
        return NeRF(D=depth, W=width, input_ch=input_ch, output_ch=5 if args.N_importance > 0 else 4, skips=[4],
                    input_ch_views=input_ch_views, input_ch_cam=getattr(args, "input_ch_cam", 0),
                    use_viewdirs=args.use_viewdirs, precision=getattr(args, "precision", "fp32"),
                    density_activation="softplus", dense_layer_init=True).to(device)
    model = network(args.netdepth, args.netwidth)
    model_fine = network(args.netdepth_fine, args.netwidth_fine) if args.N_importance > 0 else None
    _refuse_unsupported(model, model_fine)      # (at the boundary, not at the first query: render.create_nerf)
    grad_vars = list(model.parameters()) + (list(model_fine.parameters()) if model_fine is not None else [])
    box = (getattr(args, "bb_center", 0.0), getattr(args, "bb_scale", 1.0))

    def network_query_fn(inputs, viewdirs, embedded_cam, network_fn):
        return run_network(inputs, viewdirs, embedded_cam, network_fn, embed_fn=embed_fn, embeddirs_fn=embeddirs_fn,
                           bb_center=box[0], bb_scale=box[1], netchunk=getattr(args, "netchunk", 1024 * 64))
    # (optim.FlatAdam on the GPU: one launch per network's gradient buffer)
    half_range = device.type == "cuda" and getattr(args, "precision", "fp32") in L.GUARDED_PRECISIONS
    guarded = [n for n in (model, model_fine) if n is not None and n.is_supported()]      # (the layer-by-layer route is fp32: no word)
    extra = {"guards": guarded} if (half_range and guarded) else {}
    optimizer = (FlatAdam if device.type == "cuda" else torch.optim.Adam)(params=grad_vars, lr=args.lrate,
                                                                           betas=(0.9, 0.999), **extra)
    start = 0
    if not getattr(args, "no_reload", True) and os.path.isdir(os.path.join(getattr(args, "ckpt_dir", ""),
                                                                           getattr(args, "expname", ""))):
        found = [f for f in RB.checkpoint_candidates(args) if f.endswith('.tar')]
        if found:
            start = RB.restore_checkpoint(found[-1], device, model, model_fine, optimizer)
    render_kwargs_train = RB.base_render_kwargs(args, network_query_fn, model, model_fine,
                                                embedded_cam=torch.tensor((), device=device))
    render_kwargs_train.update(scene_render_params or {})
    render_kwargs_train['lindisp'] = getattr(args, "lindisp", False)
    return render_kwargs_train, RB.test_time_kwargs(render_kwargs_train, perturb=False), start, grad_vars, optimizer


class DepthTrainStep:
    """One iteration of the reference's depth-supervised loop (run_nerf_sample_based_depth.py:1126-1157):
    loss = mse(rgb) + space_carving_weight * space_carving(pred_hyp, target_h) + mse(rgb0); backward;
    clip_grad_value_(0.1); Adam.  `ray_batch` is the packed [R, 11] batch render_rays takes."""

    def __init__(self, args, render_kwargs_train, optimizer, grad_vars, distributed=None, range_check_every=100, seed=0,
                 counter_rng=True):
        """counter_rng: the step's draws (stratified jitter, importance samples, the hypotheses' u) come from a
        functional.DrawSource keyed on (seed, step, GLOBAL ray id) and are generated inside the kernels that consume
        them -- a global batch gives the same step whether one rank renders it or N ranks a shard each, like
        train.TrainStep.  False: torch.rand, as the reference draws."""
        from . import dp
        self.args, self.kw, self.optimizer, self.grad_vars = args, render_kwargs_train, optimizer, grad_vars
        self.global_step = 0
        # the 16-bit modes guard their Adam steps with the networks' range status words; the host looks every
        # `range_check_every` steps, as train.TrainStep does (0: never -- the caller does)
        self.range_check_every = int(range_check_every)
        nets = self.nets = [n for n in (self.kw["network_fn"], self.kw.get("network_fine")) if n is not None]
        distributed = torch.distributed.is_initialized() if distributed is None else distributed
        self.bucket = None
        self.rank = torch.distributed.get_rank() if distributed else 0
        self.draws = Fn.DrawSource(seed=seed) if counter_rng else None
        # both networks' backward as one launch sequence, as train.TrainStep (train.backward_merged)
        self.merged_backward = os.environ.get("PLNERF_MERGED_BWD", "1") != "0"
        self.merged_steps = 0      # steps that took it (the rest went through torch.autograd.backward)
        if distributed and torch.distributed.get_world_size() > 1:
            dp.broadcast_parameters(nets)      # replicas start from rank 0's weights (see train.TrainStep)
            dp.broadcast_optimizer_state([optimizer])
            self.bucket = dp.GradientBucket(nets)

    def check_range(self):
        """As train.TrainStep.check_range: reconcile withheld steps, raise on a set range status word."""
        withheld = self.optimizer.withheld_steps() if hasattr(self.optimizer, "withheld_steps") else 0
        for net in self.nets:
            if getattr(net, "precision", None) in L.GUARDED_PRECISIONS and net.is_supported() and \
                    next(net.parameters()).is_cuda:
                net.check_range()
        if withheld and self.bucket is not None:
            raise FloatingPointError(
                f"plnerf_amd: {withheld} optimizer step(s) were withheld because another rank's forward left the "
                "IEEE-half range (that rank's check_range() says which network)")

    def __call__(self, ray_batch, target_s, target_h, space_carving_mask=None, cached_u=None, pytest=False):
        a = self.args
        kw = {k: v for k, v in self.kw.items() if k not in ("ndc", "near", "far")}
        prev = Fn.DRAWS
        if self.draws is not None and ray_batch.is_cuda:
            self.draws.step, self.draws.ray_id0 = self.global_step, self.rank * ray_batch.shape[0]
            Fn.set_draw_source(self.draws)
        tape = Fn.MlpTape() if self.merged_backward else None
        Fn.MLP_TAPE = tape
        try:
            out = render_rays(ray_batch, retraw=True, is_joint=getattr(a, "is_joint", False), cached_u=cached_u,
                              quad_solution_v2=getattr(a, "quad_solution_v2", False), pytest=pytest, **kw)
        finally:
            Fn.set_draw_source(prev)
            Fn.MLP_TAPE = None
        self.optimizer.zero_grad()
        carve = getattr(a, "space_carving_weight", 0.) > 0. and self.global_step + 1 > getattr(a, "warm_start_nerf", 0)
        rgb, rgb0 = out['rgb_map'], out.get('rgb0')
        fused = rgb.is_cuda and rgb.dim() == 2 and rgb.shape == target_s.shape
        if fused:
            # both image terms, the space-carving term and the three gradients in one launch (plnerf_depth_loss);
            # backward((rgb, rgb0, pred_hyp), (their gradients)) is loss.backward()
            hyp = out["pred_hyp"] if carve else None
            loss5, g_rgb, g_rgb0, g_hyp = Fn.depth_loss_and_grads(
                rgb, rgb0, target_s, hyp, target_h if carve else None, getattr(a, "space_carving_weight", 0.),
                threshold=getattr(a, "space_carving_threshold", 0.0), mask=space_carving_mask if carve else None,
                is_joint=getattr(a, "is_joint", False), sharded=self.bucket is not None,
                group=self.bucket.group if self.bucket is not None else None)
            loss, img_loss, sc = loss5[0], loss5[1], loss5[3]
            roots = [(rgb, g_rgb)] + ([(rgb0, g_rgb0)] if rgb0 is not None else []) + ([(hyp, g_hyp)] if carve else [])
            from .train import backward_merged, merged_backward_ok
            if tape is not None and rgb0 is not None and merged_backward_ok(tape, self.nets):
                backward_merged(tape, self.nets, [r for r, _ in roots], [gr for _, gr in roots], self.bucket)
                self.merged_steps += 1
            else:
                torch.autograd.backward(tuple(r for r, _ in roots), tuple(gr for _, gr in roots))
        else:
            img_loss = torch.mean((rgb - target_s) ** 2)
            loss = img_loss
            sc = torch.zeros((), device=ray_batch.device)
            if carve:
                if getattr(a, "is_joint", False) and self.bucket is not None:
                    # (is_joint chooses the hypothesis from the mean over the WHOLE batch: a rank's shard alone would choose
                    # its own -- the fused path adds the shards' column sums first, functional.joint_choice)
                    raise NotImplementedError("plnerf_amd: is_joint=True under data parallelism needs the fused loss "
                                              "(CUDA tensors, rgb [R, 3]); this call took the torch expressions")
                sc = compute_space_carving_loss(out["pred_hyp"], target_h, is_joint=getattr(a, "is_joint", False),
                                                norm_p=getattr(a, "norm_p", 2),
                                                threshold=getattr(a, "space_carving_threshold", 0.0),
                                                mask=space_carving_mask)
                loss = loss + a.space_carving_weight * sc
            if rgb0 is not None:
                loss = loss + torch.mean((rgb0 - target_s) ** 2)
            loss.backward()
        flat_adam = isinstance(self.optimizer, FlatAdam) and all(fl is not None for fl in self.optimizer._flat)
        scale, tails = 1.0, None
        if self.bucket is not None:
            # one wait for the two collectives; the 1 / world factor and the guard (the ranks' summed range status, which
            # travelled behind the gradients) go into the step kernel (dp.GradientBucket)
            scale = self.bucket.finish(defer_scale=flat_adam)
            tails = self.bucket.tails()
            tails = tails if len(tails) == len(self.nets) else None
        if flat_adam:
            # clip_grad_value_(0.1) folded into the step kernel (:1156), applied to the averaged gradient
            self.optimizer.step(clip_value=0.1, grad_scale=scale, guards=tails)
        elif isinstance(self.optimizer, FlatAdam):
            self.optimizer.step(clip_value=0.1)
        else:
            torch.nn.utils.clip_grad_value_(self.grad_vars, 0.1)
            self.optimizer.step()
        self.global_step += 1
        if self.range_check_every and self.global_step % self.range_check_every == 0:
            self.check_range()
        return loss.detach(), img_loss.detach(), sc.detach(), out
