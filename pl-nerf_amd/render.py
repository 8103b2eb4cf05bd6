"""The render operator surface of run_plnerf.py, on the HIP path.

Same names, arguments and return structures as the reference (file:line cited per
function), so a caller of the reference's `create_nerf` / `render` / `render_rays` can
switch to this module unchanged.  Everything numerical happens in libplnerf_hip.so via
`functional.py`; torch is used for allocation, the random draws and the autograd tape.
"""
import numpy as np
import torch

from . import _lib as L
from . import functional as Fn
from . import raybatch as RB
from .nerf import NeRF, Embedder, get_embedder
from .optim import FlatAdam
from .rays import get_rays, ndc_rays

DEBUG = False
# Tests only (tests/test_gpu_fullsize.py): a dict that render_rays fills with its intermediate tensors -- coarse depths, raw,
# weights, tau, T, the importance samples with their search indices, the merged depths, the final weights.  While it is set
# the coarse pass's epilogue runs as its separate launches (bit-identical to the fused one:
# test_fused_coarse_epilogue_equals_separate_launches), which is where those tensors exist in HBM.
STAGE_TAP = None
MAX_ROWS_PER_LAUNCH = 1 << 21   # MLP rows per kernel launch when activations are saved (~21 GB fp32)


def batchify(fn, chunk):
    """run_plnerf.py:68-75: apply `fn` to row blocks of at most `chunk` rows (None = all at once)."""
    if chunk is None:
        return fn
    return lambda inputs: torch.cat([fn(block) for block in torch.split(inputs, chunk, dim=0)], 0)


def _fusable(fn, embed_fn, embeddirs_fn, viewdirs):
    # the kernel's own encoding: 3 + 6 L | 3 + 6 M channels (multires L <= 10, multires_views M <= 4), the reference's
    # Embedder with the matching number of frequencies
    if not (isinstance(fn, NeRF) and fn.is_supported() and fn.has_fused_encoding() and
            isinstance(embed_fn, Embedder) and embed_fn.is_standard((fn.input_ch - 3) // 6)):
        return False
    if not fn.use_viewdirs:      # (output_linear on the trunk: the view directions, if any were passed, are not used)
        return True
    return viewdirs is not None and isinstance(embeddirs_fn, Embedder) and \
        embeddirs_fn.is_standard((fn.input_ch_views - 3) // 6)


def _hip_embedding(inputs, viewdirs, embed_fn, embeddirs_fn):
    """The reference's Embedder pair at ANY frequency count up to 16 (multires 11..16 included: outside the fused kernel's
    in-register encoding) as one plnerf_embed_rows launch: [R * S, 3 + 6 L (+ 3 + 6 M)] rows, the view direction encoded per
    sample as run_plnerf.py:85-88 does.  None when the encoders are not the standard ones or the rays need a gradient."""
    if not (inputs.is_cuda and inputs.dim() == 3 and inputs.shape[-1] == 3 and not inputs.requires_grad and
            isinstance(embed_fn, Embedder) and embed_fn.is_standard(embed_fn.num_freqs) and 0 <= embed_fn.num_freqs <= 16):
        return None
    if viewdirs is None:
        return Fn.embed_rows(inputs, None, None, embed_fn.num_freqs, 0)
    if not (isinstance(embeddirs_fn, Embedder) and embeddirs_fn.is_standard(embeddirs_fn.num_freqs) and
            0 <= embeddirs_fn.num_freqs <= 16 and not viewdirs.requires_grad):
        return None
    return Fn.embed_rows(inputs, viewdirs, None, embed_fn.num_freqs, embeddirs_fn.num_freqs)


def run_network(inputs, viewdirs, fn, embed_fn, embeddirs_fn, netchunk=1024 * 64):
    """run_plnerf.py:78-92: encode sample positions / view directions and apply the MLP.

    Hot path (the reference's default encoders + network): one fused kernel per launch --
    encoding in the prologue, direction encoded per ray.  `netchunk` "does not affect final
    results" in the reference; here rows are only split to bound the saved-activation
    buffer.  Any other encoder/network combination takes the generic route below and still
    ends in the HIP MLP through NeRF.forward."""
    if _fusable(fn, embed_fn, embeddirs_fn, viewdirs):
        R, S = inputs.shape[0], inputs.shape[1]
        rays_per_launch = max(1, MAX_ROWS_PER_LAUNCH // max(S, 1))
        if not fn.use_viewdirs:
            viewdirs = None
        if R <= rays_per_launch or not torch.is_grad_enabled():      # (inference saves nothing: one launch whatever the size)
            return fn.query(inputs, viewdirs)
        # equal shares: a remainder launch of a few hundred rows costs a whole tile walk (60 us for 384 rows of a 32,768-ray
        # chunk at 192 samples, 57 times per 800 x 800 frame -- profiles/r05_render_frame_kernel_stats.csv)
        n_launches = -(-R // rays_per_launch)
        per = -(-R // n_launches)
        outs = [fn.query(inputs[i:i + per], None if viewdirs is None else viewdirs[i:i + per]) for i in range(0, R, per)]
        return torch.cat(outs, 0)
    embedded = _hip_embedding(inputs, viewdirs, embed_fn, embeddirs_fn)
    if embedded is None:      # (another encoder: its own torch expressions)
        inputs_flat = torch.reshape(inputs, [-1, inputs.shape[-1]])
        embedded = embed_fn(inputs_flat)
        if viewdirs is not None:
            input_dirs = viewdirs[:, None].expand(inputs.shape)
            embedded = torch.cat([embedded, embeddirs_fn(torch.reshape(input_dirs, [-1, input_dirs.shape[-1]]))], -1)
    if isinstance(fn, NeRF) and fn.is_supported() and embedded.is_cuda:
        netchunk = max(int(netchunk), MAX_ROWS_PER_LAUNCH)      # rows are independent: fewer, larger launches
    outputs_flat = batchify(fn, netchunk)(embedded)
    return torch.reshape(outputs_flat, list(inputs.shape[:-1]) + [outputs_flat.shape[-1]])


def compute_weights(raw, z_vals, rays_d, noise=0.):
    """run_plnerf.py:504-513 (piecewise-constant opacity)."""
    near = z_vals[..., :1]
    noise_t = _noise_tensor(noise, raw)
    return Fn.QuadratureFn.apply(_rgb_sigma(raw), z_vals, near, near, rays_d, noise_t, "constant", "midpoint", False,
                                 False)[3]


def compute_weights_piecewise_linear(raw, z_vals, near, far, rays_d, noise=0., return_tau=False):
    """run_plnerf.py:516-550."""
    noise_t = _noise_tensor(noise, raw)
    out = Fn.QuadratureFn.apply(_rgb_sigma(raw), z_vals, near, far, rays_d, noise_t, "linear", "midpoint", False, False)
    if return_tau:
        return out[3], out[5], out[6]
    return out[3]


def _rgb_sigma(raw):
    """The reference reads channels 0..3 of raw (rgb, sigma) and ignores any further ones (run_plnerf.py:566-570);
    the kernels take exactly four.  (A slice here stays on the autograd tape.)"""
    return raw if raw.shape[-1] == 4 else raw[..., :4]


def _draw_noise(raw, raw_noise_std, pytest):
    """The density noise of raw2outputs (run_plnerf.py:568-576), or None."""
    if not raw_noise_std > 0.:
        return None
    shape = list(raw.shape[:-1])
    if pytest:   # the reference's deterministic draw is UNIFORM (run_plnerf.py:573-576)
        return Fn.numpy_uniform(shape, raw.device) * raw_noise_std
    if Fn.DRAWS is not None and len(shape) == 2 and raw.is_cuda:
        # counter-based normal draws keyed on the global ray id: like the jitter and the sampler's u, the density noise of
        # the LLFF configurations does not depend on how the batch is sharded
        return Fn.DRAWS.normal(shape[0], shape[1], Fn.DRAWS.next_noise_stream(), raw.device) * raw_noise_std
    return torch.randn(shape, device=raw.device) * raw_noise_std


def _noise_tensor(noise, raw):
    if isinstance(noise, torch.Tensor):
        return noise.to(raw.device)
    if noise == 0.:
        return None
    return torch.full(raw.shape[:-1], float(noise), device=raw.device)


def raw2outputs(raw, z_vals, near, far, rays_d, mode, color_mode, raw_noise_std=0, pytest=False,
                white_bkgd=False, farcolorfix=False):
    """run_plnerf.py:553-624.  Returns (rgb_map, disp_map, acc_map, weights, depth_map, tau, T)
    with tau = T = None in constant mode."""
    if mode not in ("linear", "constant"):
        raise ValueError(f"mode must be 'linear' or 'constant', got {mode!r}")
    if mode == "linear" and color_mode not in ("midpoint", "left"):
        raise ValueError("Color mode unimplemented, please select left or midpoint.")
    noise = _draw_noise(raw, raw_noise_std, pytest)
    cm = color_mode if mode == "linear" else "midpoint"
    rgb, disp, acc, w, depth, tau, T = Fn.QuadratureFn.apply(_rgb_sigma(raw), z_vals, near, far, rays_d, noise, mode,
                                                             cm, white_bkgd, farcolorfix)
    if mode == "constant":
        tau, T = None, None
    return rgb, disp, acc, w, depth, tau, T


def _draw_u(prefix_shape, n, det, pytest, device):
    if pytest:   # run_nerf_helpers.py:256-264
        if det:
            return torch.Tensor(np.linspace(0., 1., n)).to(device)
        return Fn.numpy_uniform(list(prefix_shape) + [n], device)
    if det:
        return Fn.cpu_linspace(n, device)
    if Fn.DRAWS is not None and len(prefix_shape) == 1:
        # an installed DrawSource supplies EVERY draw of the step (the constant-mode sampler and the `constant_init`
        # warm-up included): counter-based on the global ray id, so the step does not depend on the sharding
        return Fn.DRAWS.uniform(int(prefix_shape[0]), n, Fn.DrawSource.U, device)
    return torch.rand(list(prefix_shape) + [n], device=device)


def sample_pdf(bins, weights, N_samples, det=False, pytest=False):
    """run_nerf_helpers.py:241-284."""
    u = _draw_u(bins.shape[:-1], N_samples, det, pytest, bins.device)
    return Fn.sample_const(bins, weights, u)


def sample_pdf_reformulation(bins, weights, tau, T, near, far, N_samples, det=False, pytest=False,
                             quad_solution_v2=False, zero_threshold=1e-4, epsilon_=1e-3):
    """run_nerf_helpers.py:364-445.  Returns (samples, T_below, tau_below, bin_below).
    quad_solution_v2 is accepted and ignored, as in the reference."""
    u = _draw_u(bins.shape[:-1], N_samples, det, pytest, bins.device)
    return Fn.sample_pl(bins, weights, tau, T, near, far, u, zero_threshold, epsilon_, want_extras=True)


def render_rays(ray_batch, network_fn, network_query_fn, N_samples, mode, color_mode, retraw=False,
                lindisp=False, perturb=0., N_importance=0, network_fine=None, white_bkgd=False,
                raw_noise_std=0., verbose=False, pytest=False, quad_solution_v2=False, zero_tol=1e-4,
                epsilon=1e-3, farcolorfix=False, constant_init=False):
    """Volumetric rendering of a ray batch -- run_plnerf.py:627-758.

    ray_batch [R, 8 or 11] = origin(3) direction(3) near far [unit view direction(3)].
    Returns the reference's dict: rgb_map, disp_map, acc_map, depth_map (+ raw if retraw;
    + rgb0, disp0, depth0, acc0, z_std if N_importance > 0)."""
    dev = ray_batch.device
    N_rays = ray_batch.shape[0]
    if isinstance(ray_batch, RB.RayColumns):
        rays_o, rays_d, near, far, viewdirs = (ray_batch.rays_o, ray_batch.rays_d, ray_batch.near.reshape(-1, 1),
                                               ray_batch.far.reshape(-1, 1), ray_batch.viewdirs)
    else:
        # one contiguous copy of each column group: every kernel below takes them as they are (the slices of the
        # packed batch would otherwise be re-copied by each launch that consumes them)
        rays_o, rays_d = ray_batch[:, 0:3].contiguous(), ray_batch[:, 3:6].contiguous()
        viewdirs = ray_batch[:, -3:].contiguous() if ray_batch.shape[-1] > 8 else None
        near, far = ray_batch[:, 6:7].contiguous(), ray_batch[:, 7:8].contiguous()
    # A ray batch that requires a gradient (no reference training path has one; camera-pose optimisation would): autograd
    # carries d(loss)/d(rays) through z_vals, the sample positions, the interval lengths and the MLP's inputs
    # (run_plnerf.py:683-707, 731-735 are torch expressions there).  The same expressions run here as torch operations, the
    # quadrature returns the geometry's gradient (plnerf_quad_bwd_rays) and the MLP its inputs' (plnerf_mlp_input_grad);
    # the fused position / epilogue kernels are for batches without one.
    rays_grad = torch.is_grad_enabled() and any(
        t is not None and t.requires_grad for t in (rays_o, rays_d, near, far, viewdirs))

    t_vals = Fn.cpu_linspace(N_samples, dev)
    # Without a gradient for the ray batch the prologue's element-wise chain runs as ONE kernel (depths + positions,
    # bit-identical to the torch expressions below, which remain for an empty batch and for rays_grad).
    fused_glue = ray_batch.is_cuda and N_rays > 0 and not rays_grad
    # Random draws: pytest=True replays the reference's numpy draws; otherwise an installed functional.DrawSource
    # supplies counter-based draws (inside the consuming kernels on the fused path), else torch.rand as the
    # reference does.
    draws = None if pytest else Fn.DRAWS
    if draws is not None:
        draws.noise_calls = 0      # (this call's coarse pass draws its density noise first, then the fine pass)
    if fused_glue:
        t_rand = None
        if perturb > 0. and (pytest or draws is None):
            shape = [N_rays, N_samples]
            t_rand = Fn.numpy_uniform(shape, dev) if pytest else torch.rand(shape, device=dev)
        z_vals, pts = Fn.coarse_samples(rays_o, rays_d, near, far, t_vals, t_rand, lindisp, perturb > 0., draws)
    else:
        if not lindisp:
            z_vals = near * (1. - t_vals) + far * t_vals
        else:
            z_vals = 1. / (1. / near * (1. - t_vals) + 1. / far * t_vals)
        z_vals = z_vals.expand([N_rays, N_samples])

        if perturb > 0.:
            mids = .5 * (z_vals[..., 1:] + z_vals[..., :-1])
            upper = torch.cat([mids, z_vals[..., -1:]], -1)
            lower = torch.cat([z_vals[..., :1], mids], -1)
            if pytest:
                t_rand = Fn.numpy_uniform(list(z_vals.shape), dev)
            elif draws is not None:
                t_rand = draws.uniform(N_rays, N_samples, Fn.DrawSource.T_RAND, dev)
            else:
                t_rand = torch.rand(z_vals.shape, device=dev)
            z_vals = lower + (upper - lower) * t_rand

        pts = rays_o[..., None, :] + rays_d[..., None, :] * z_vals[..., :, None]

    if constant_init:   # run_plnerf.py:710-711: overrides the mode for the whole call
        mode = "constant"

    raw = network_query_fn(pts, viewdirs, network_fn)
    tap = STAGE_TAP
    fused_epilogue = fused_glue and N_importance > 0 and mode == "linear" and color_mode in ("midpoint", "left") \
        and tap is None
    if tap is not None:
        tap.update(z_vals0=z_vals, raw0=raw)
    if fused_epilogue:
        # coarse raw2outputs + sampler + clamp + sort + fine positions + z_std: one launch (weights, tau, T and the
        # cdf never reach HBM); identical values to the separate calls below
        det = perturb == 0.
        u = _draw_u([N_rays], N_importance, det, pytest, dev) if (pytest or det or draws is None) else None
        rgb_map_0, disp_map_0, acc_map_0, depth_map_0, z_vals, pts, z_std = Fn.CoarseEpilogueFn.apply(
            _rgb_sigma(raw), z_vals, near, far, rays_o, rays_d, _draw_noise(raw, raw_noise_std, pytest), u,
            N_importance, color_mode, white_bkgd, farcolorfix, zero_tol, epsilon, draws, False)
    else:
        rgb_map, disp_map, acc_map, weights, depth_map, tau, T = raw2outputs(
            raw, z_vals, near, far, rays_d, mode, color_mode, raw_noise_std, pytest=pytest, white_bkgd=white_bkgd,
            farcolorfix=farcolorfix)

    if N_importance > 0 and not fused_epilogue:
        rgb_map_0, disp_map_0, acc_map_0, depth_map_0 = rgb_map, disp_map, acc_map, depth_map
        with torch.set_grad_enabled(torch.is_grad_enabled() and not rays_grad):      # (rays_grad: detached below either way)
            if mode == "linear" and tap is not None:
                u = _draw_u(z_vals.shape[:-1], N_importance, perturb == 0., pytest, dev)
                z_samples, inds = Fn.sample_pl(z_vals, weights, tau, T, near, far, u, zero_tol, epsilon, want_inds=True)
                tap.update(weights0=weights, tau0=tau, T0=T, u=u, inds=inds)
            elif mode == "linear":
                z_samples, _, _, _ = sample_pdf_reformulation(
                    z_vals, weights, tau, T, near, far, N_importance, det=(perturb == 0.), pytest=pytest,
                    quad_solution_v2=quad_solution_v2, zero_threshold=zero_tol, epsilon_=epsilon)
            elif mode == "constant":
                z_vals_mid = .5 * (z_vals[..., 1:] + z_vals[..., :-1])
                z_samples = sample_pdf(z_vals_mid, weights[..., 1:-1], N_importance, det=(perturb == 0.),
                                       pytest=pytest)
        z_samples = z_samples.detach()
        if rays_grad:
            # run_plnerf.py:731-734 as written: the clamp's bounds and the coarse depths carry the batch's gradient
            z_clamped = torch.clamp(z_samples, near, far)
            z_vals, order = torch.sort(torch.cat([z_vals, z_clamped], -1), -1)
            z_std = torch.std(z_clamped, dim=-1, unbiased=False)
            if tap is not None:
                # (a fine sample ON a coarse depth -- the sampler returns the left knot where the density is flat,
                # run_nerf_helpers.py:425 -- ties with it, and the tied slots' gradients
                # reach near / far through whichever of the two the sort placed there -- torch.sort is not stable, here as
                # in the reference; a test that compares gradients slot by slot needs the order that was used)
                tap.update(sort_order=order)
        else:
            # clamp + cat + sort in one kernel; z_samples clamped for z_std
            z_vals = Fn.merge_sort(z_vals, z_samples, near, far)
            z_std = torch.std(torch.clamp(z_samples, near, far), dim=-1, unbiased=False)
        if fused_glue:
            pts = Fn.ray_points(rays_o, rays_d, z_vals)
        else:
            pts = rays_o[..., None, :] + rays_d[..., None, :] * z_vals[..., :, None]
        if tap is not None:
            tap.update(z_samples=z_samples, z_fine=z_vals)

    if N_importance > 0:
        run_fn = network_fn if network_fine is None else network_fine
        raw = network_query_fn(pts, viewdirs, run_fn)
        rgb_map, disp_map, acc_map, weights, depth_map, tau, T = raw2outputs(
            raw, z_vals, near, far, rays_d, mode, color_mode, raw_noise_std, pytest=pytest,
            white_bkgd=white_bkgd, farcolorfix=farcolorfix)
        if tap is not None:
            tap.update(weights=weights)

    ret = {'rgb_map': rgb_map, 'disp_map': disp_map, 'acc_map': acc_map, 'depth_map': depth_map}
    if retraw:
        ret['raw'] = raw
    if N_importance > 0:
        ret['rgb0'] = rgb_map_0
        ret['disp0'] = disp_map_0
        ret['depth0'] = depth_map_0
        ret['acc0'] = acc_map_0
        ret['z_std'] = z_std

    if DEBUG:
        for k in ret:
            if torch.isnan(ret[k]).any() or torch.isinf(ret[k]).any():
                print(f"! [Numerical Error] {k} contains nan or inf.")
    return ret


def batchify_rays(rays_flat, chunk=1024 * 32, **kwargs):
    """run_plnerf.py:95-107: render_rays over chunks of the flat ray rows."""
    def one_chunk(rows, first_row):
        if Fn.DRAWS is not None:
            Fn.DRAWS.chunk_offset = first_row       # draws are keyed on the ray's position in the whole batch
        return render_rays(rows, **kwargs)
    try:
        return RB.map_row_chunks(one_chunk, rays_flat, chunk)
    finally:
        if Fn.DRAWS is not None:
            Fn.DRAWS.chunk_offset = 0


def render(H, W, K, chunk=1024 * 32, rays=None, c2w=None, ndc=True, near=0., far=1., use_viewdirs=False,
           c2w_staticcam=None, **kwargs):
    """run_plnerf.py:110-175.  Rays come from `rays` = (origins, directions) or from a full view `c2w`; returns
    [rgb_map, disp_map, acc_map, extras] shaped like the ray arrays."""
    rays_o, rays_d = get_rays(H, W, K, c2w) if c2w is not None else rays
    extra = []
    if use_viewdirs:
        extra.append(RB.unit_directions(rays_d))                # of the ORIGINAL directions (:146-150) ...
        if c2w_staticcam is not None:                            # ... while the geometry may come from a fixed camera
            rays_o, rays_d = get_rays(H, W, K, c2w_staticcam)
    lead_shape = rays_d.shape[:-1]
    if ndc:                                                       # forward-facing scenes (:153-155)
        rays_o, rays_d = ndc_rays(H, W, K[0][0], 1., rays_o, rays_d)
    rows, _ = RB.pack_rays(rays_o, rays_d, near, far, extra)
    return RB.unflatten_outputs(batchify_rays(rows, chunk, **kwargs), tuple(lead_shape))


def render_path(render_poses, hwf, K, chunk, render_kwargs, gt_imgs=None, savedir=None, render_factor=0):
    """run_plnerf.py:178-216: render() once per pose; returns (rgbs [n,H,W,3], disps [n,H,W]) as numpy arrays.
    Frames are rendered under torch.no_grad() (the reference calls this inside `with torch.no_grad()`); writing
    PNGs (`savedir`) is the caller's business here -- image IO is outside the path -- so a
    non-None savedir raises instead of being silently ignored."""
    if savedir is not None:
        raise NotImplementedError("render_path does not write images; save the returned arrays on the caller side")
    H, W, focal = hwf
    if render_factor != 0:
        H, W, focal = H // render_factor, W // render_factor, focal / render_factor
    rgbs, disps = [], []
    with torch.no_grad():
        for c2w in render_poses:
            rgb, disp, acc, _ = render(H, W, K, chunk=chunk, c2w=c2w[:3, :4], **render_kwargs)
            rgbs.append(rgb.cpu().numpy())
            disps.append(disp.cpu().numpy())
            for net in (render_kwargs.get('network_fn'), render_kwargs.get('network_fine')):
                if isinstance(net, NeRF) and net.precision in L.GUARDED_PRECISIONS and net.is_supported():
                    net.check_range()       # (the frame was just synchronised) a clamped frame must not pass silently
    return np.stack(rgbs, 0), np.stack(disps, 0)


def _refuse_unsupported(*nets):
    """create_nerf's boundary check.  Until round 5 a flag combination the compiled trunk cannot express (netdepth > 8,
    netwidth > 256, several live skips, multires > 10 -- INTEGRATION.md has the table) raised here; since round 6 such a
    network is SERVED, layer by layer on exact-fp32 MFMA products (generic.py) -- several times slower per row than the fused
    kernels and fp32 whatever `precision` says, which the caller is told once, here, before any data is loaded.  What still
    raises: a shape the reference's own forward cannot run either (a skip after the last trunk layer)."""
    import warnings
    for net in nets:
        if net is None or net.is_supported():
            continue
        if any(k == net.D - 1 for k in net.skips):
            raise NotImplementedError(f"skips={net.skips} concatenates after the last trunk layer (netdepth {net.D}): the "
                                      "reference's head layers cannot consume that either (run_nerf_helpers.py:109-116)")
        warnings.warn(f"plnerf_amd: D={net.D}, W={net.W}, skips={net.skips}, input_ch={net.input_ch}, input_ch_views="
                      f"{net.input_ch_views} is outside the fused kernels' trunk: it runs layer by layer on exact-fp32 "
                      "products (generic.py), unfused and fp32 in every precision mode", RuntimeWarning, stacklevel=3)


def create_nerf(args, device=None):
    """run_plnerf.py:417-502: (render_kwargs_train, render_kwargs_test, start, grad_vars, optimizer,
    optimizer_coarse).  `device` (extension) defaults to cuda:<current>; optional `args.precision` picks the MLP
    arithmetic ("fp32" default).

    `optimizer` drives the fine network and is the one a checkpoint restores; `optimizer_coarse` the coarse network
    (:438, :446-447).  Both are the reference's Adam (lr, betas (0.9, 0.999)): optim.FlatAdam on the GPU -- a
    torch.optim.Adam with the same state_dict that steps a network in one plnerf_adam_step launch."""
    device = RB.default_device(device)
    precision = getattr(args, "precision", "fp32")
    embed_fn, input_ch = get_embedder(args.multires, args.i_embed)
    embeddirs_fn, input_ch_views = (get_embedder(args.multires_views, args.i_embed) if args.use_viewdirs
                                    else (None, 0))

    def network(depth, width):
        return NeRF(D=depth, W=width, input_ch=input_ch, output_ch=5 if args.N_importance > 0 else 4, skips=[4],
                    input_ch_views=input_ch_views, use_viewdirs=args.use_viewdirs, precision=precision).to(device)
    model = network(args.netdepth, args.netwidth)
    model_fine = network(args.netdepth_fine, args.netwidth_fine) if args.N_importance > 0 else None
    _refuse_unsupported(model, model_fine)
    coarse_vars = list(model.parameters())
    grad_vars = coarse_vars if model_fine is None else list(model_fine.parameters())

    def network_query_fn(inputs, viewdirs, network_fn):
        return run_network(inputs, viewdirs, network_fn, embed_fn=embed_fn, embeddirs_fn=embeddirs_fn,
                           netchunk=args.netchunk)

    on_gpu = device.type == "cuda"
    # modes whose kernels can clamp at the IEEE-half maximum (forward: f16x3 / f16; saved planes: every 16-bit mode):
    # the steps are guarded by the network's range status word
    half_range = on_gpu and precision in L.GUARDED_PRECISIONS

    def guard(*nets):
        nets = [n for n in nets if n is not None and n.is_supported()]
        return {"guards": nets} if (half_range and nets) else {}
    adam = FlatAdam if on_gpu else torch.optim.Adam
    # (the fine network's step is guarded by BOTH networks: its samples and z_vals come from the coarse pass, and a
    # clamped coarse forward must withhold the whole step, not let the two networks drift apart until the next poll)
    optimizer = adam(params=grad_vars, lr=args.lrate, betas=(0.9, 0.999),
                     **(guard(model_fine, model) if model_fine is not None else guard(model)))
    # Single-pass configuration (N_importance == 0): the reference builds BOTH Adams over the same (coarse)
    # parameters and steps them one after the other.  The second FlatAdam adopts the flat buffer the first one
    # re-homed the weights into (optim.FlatAdam._flatten) and carries the same guard.
    optimizer_coarse = adam(params=coarse_vars, lr=args.coarse_lrate, betas=(0.9, 0.999), **guard(model))

    start = 0
    candidates = RB.checkpoint_candidates(args)
    if candidates and not args.no_reload:
        start = RB.restore_checkpoint(candidates[-1], device, model, model_fine, optimizer)

    render_kwargs_train = RB.base_render_kwargs(args, network_query_fn, model, model_fine)
    if args.dataset != 'llff' or args.no_ndc:      # NDC only for forward-facing LLFF data (:490-493)
        render_kwargs_train['ndc'] = False
        render_kwargs_train['lindisp'] = args.lindisp
    render_kwargs_test = RB.test_time_kwargs(render_kwargs_train, perturb=True)
    return render_kwargs_train, render_kwargs_test, start, grad_vars, optimizer, optimizer_coarse
