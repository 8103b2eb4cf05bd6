"""Host plumbing shared by the two render surfaces (render.py = run_plnerf.py, depth.py = the depth-supervised
script): packing rays into the flat rows the kernels consume, running a per-row function over bounded chunks,
shaping the result dict back, and the pieces of `create_nerf` both variants need (network construction from the
argument namespace, checkpoint discovery and restore, the render-kwargs dict).

Nothing numerical happens here; the reference's behaviour these helpers reproduce is cited where it is used.
"""
import os

import torch

HEAD_KEYS = ("rgb_map", "disp_map", "acc_map")     # what render() returns positionally (run_plnerf.py:172-175)


def unit_directions(d):
    """Unit view directions as flat fp32 rows (run_plnerf.py:148-150)."""
    return (d / torch.linalg.vector_norm(d, dim=-1, keepdim=True)).reshape(-1, 3).float()


def pack_rays(rays_o, rays_d, near, far, extra_columns=()):
    """[R, 8 + ...] rows = origin(3) direction(3) near far [extras...], fp32 (run_plnerf.py:158-164).  `near` / `far`
    are scalars broadcast over the rays.  Returns (rows, leading shape of the ray arrays)."""
    lead = tuple(rays_d.shape[:-1])
    o = rays_o.reshape(-1, 3).float()
    d = rays_d.reshape(-1, 3).float()
    n = d.shape[0]
    bounds = torch.empty(n, 2, device=d.device, dtype=torch.float32)
    bounds[:, 0] = near
    bounds[:, 1] = far
    return torch.cat([o, d, bounds] + [c.reshape(n, -1).float() for c in extra_columns], -1), lead


class RayColumns:
    """A ray batch held as separate contiguous columns (what plnerf_select_rays writes) instead of the packed
    [R, 11] rows: render_rays takes either; this form saves the five slice-copies per call."""

    def __init__(self, rays_o, rays_d, near, far, viewdirs=None):
        self.rays_o, self.rays_d, self.near, self.far, self.viewdirs = rays_o, rays_d, near, far, viewdirs
        self.shape = (rays_o.shape[0], 11 if viewdirs is not None else 8)
        self.device, self.is_cuda, self.requires_grad = rays_o.device, rays_o.is_cuda, False

    def packed(self):
        cols = [self.rays_o, self.rays_d, self.near.reshape(-1, 1), self.far.reshape(-1, 1)]
        return torch.cat(cols + ([self.viewdirs] if self.viewdirs is not None else []), -1)


def map_row_chunks(fn, rows, chunk):
    """fn(rows[i:i+chunk], i) -> dict of tensors, over consecutive chunks; values concatenated along dim 0 (what
    the reference's batchify_rays does to bound memory, run_plnerf.py:95-107).  The second argument is the chunk's
    first row, for draws keyed on the ray's position in the batch."""
    if rows.shape[0] <= chunk:
        return dict(fn(rows, 0))
    parts = [fn(rows[i:i + chunk], i) for i in range(0, rows.shape[0], chunk)]
    return {k: torch.cat([p[k] for p in parts], 0) for k in parts[0]}


def unflatten_outputs(outputs, lead):
    """Per-ray outputs [R, ...] -> [*lead, ...]; returns [rgb_map, disp_map, acc_map, {the rest}]."""
    shaped = {k: v.reshape(*lead, *v.shape[1:]) for k, v in outputs.items()}
    return [shaped[k] for k in HEAD_KEYS] + [{k: v for k, v in shaped.items() if k not in HEAD_KEYS}]


def default_device(device=None):
    if device is not None:
        return torch.device(device)
    return torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else torch.device("cpu")


def checkpoint_candidates(args):
    """The files create_nerf may resume from: args.ft_path if given, else every name containing 'tar' under
    ckpt_dir/expname, sorted (run_plnerf.py:454-457; the newest is the last)."""
    ft = getattr(args, "ft_path", None)
    if ft is not None and ft != 'None':
        return [ft]
    ckdir = os.path.join(getattr(args, "ckpt_dir", ""), getattr(args, "expname", ""))
    return [os.path.join(ckdir, f) for f in sorted(os.listdir(ckdir)) if 'tar' in f]


def restore_checkpoint(path, device, model, model_fine, optimizer):
    """Load the reference's checkpoint dict (run_plnerf.py:1324-1332) into the given objects; returns global_step.
    Only the fine optimizer's state is stored there, so `optimizer` is the one create_nerf hands back first."""
    ckpt = torch.load(path, map_location=device)
    optimizer.load_state_dict(ckpt['optimizer_state_dict'])
    model.load_state_dict(ckpt['network_fn_state_dict'])
    if model_fine is not None:
        model_fine.load_state_dict(ckpt['network_fine_state_dict'])
    return ckpt['global_step']


def base_render_kwargs(args, network_query_fn, model, model_fine, **extra):
    """The keys both scripts put into render_kwargs_train (run_plnerf.py:475-488)."""
    kw = dict(network_query_fn=network_query_fn, perturb=args.perturb, N_importance=args.N_importance,
              network_fine=model_fine, N_samples=args.N_samples, network_fn=model, use_viewdirs=args.use_viewdirs,
              white_bkgd=args.white_bkgd, raw_noise_std=args.raw_noise_std, mode=args.mode,
              color_mode=args.color_mode)
    kw.update(extra)
    return kw


def test_time_kwargs(train_kwargs, perturb):
    """render_kwargs_test: a copy with the scripts' test-time overrides (run_plnerf.py:497-499: perturb True;
    the depth script: False)."""
    kw = dict(train_kwargs)
    kw['perturb'] = perturb
    kw['raw_noise_std'] = 0.
    return kw
