"""Measurement support shared by bench.py and tools/: an ANALYTIC scene a NeRF can fit (there is no dataset in the
image) and the "PSNR vs ref" run on it -- BASELINE.json's metric is "training rays/sec ...; PSNR vs ref", and the
reference prints the training PSNR every step (run_plnerf.py:1290-1296).

The scene: a unit sphere at the origin, surface colour 0.5 + 0.5 p (a smooth function of the surface point), white
background -- rendered in closed form for the views of the NeRF-synthetic camera ring (pose_spherical(theta, -30, 4),
800 x 800, focal from camera_angle_x = 0.6911112: load_blender.py:99-102) into images that live on the device, which is
what the reference's loader would have put there.  Nothing here is on the timed hot path.
"""
import math
import os
import sys
import tempfile
from argparse import Namespace

import torch


def blender_intrinsics(H=800, W=800):
    focal = .5 * W / math.tan(.5 * 0.6911112070083618)
    return [[focal, 0, .5 * W], [0, focal, .5 * H], [0, 0, 1]]


def analytic_image(P, H, W, K, c2w, dev):
    """The sphere scene seen from `c2w`, [H, W, 3] on `dev` (closed form: first ray / sphere intersection)."""
    o, d = P.get_rays(H, W, K, c2w.to(dev))
    dn = d / d.norm(dim=-1, keepdim=True)
    b = (o * dn).sum(-1)
    disc = b * b - ((o * o).sum(-1) - 1.0)
    t = -b - torch.sqrt(torch.clamp(disc, min=0))
    p = o + dn * t[..., None]
    return torch.where((disc > 0)[..., None], 0.5 + 0.5 * p, torch.ones_like(p)).contiguous()


class AnalyticScene:
    """`n_views` training views on the camera ring and one held-out view between two of them."""

    def __init__(self, P, n_views, dev, H=800, W=800, heldout_hw=200):
        self.H, self.W, self.K = H, W, blender_intrinsics(H, W)
        self.near, self.far = 2.0, 6.0
        self.poses = [P.rays.pose_spherical(-180.0 + 360.0 * i / n_views, -30.0, 4.0)[:3, :4] for i in range(n_views)]
        self.images = [analytic_image(P, H, W, self.K, c2w, dev) for c2w in self.poses]
        # held out: a pose the training ring does not contain, at a size a no-grad render finishes in ~25 ms
        self.h_hw = heldout_hw
        self.h_K = blender_intrinsics(heldout_hw, heldout_hw)
        self.h_pose = P.rays.pose_spherical(-180.0 + 180.0 / n_views, -30.0, 4.0)[:3, :4].to(dev)
        self.h_image = analytic_image(P, heldout_hw, heldout_hw, self.h_K, self.h_pose, dev)


def nerf_args(precision, ckpt_dir, n_samples=64, n_importance=128, n_rand=4096, lrate_decay=500):
    """BASELINE configs[1]'s flags (configs/blender_linear.txt with 64 + 128 samples)."""
    return Namespace(multires=10, i_embed=0, use_viewdirs=True, multires_views=4, N_importance=n_importance,
                     N_samples=n_samples, netdepth=8, netwidth=256, netdepth_fine=8, netwidth_fine=256, netchunk=65536,
                     lrate=5e-4, coarse_lrate=5e-4, ft_path=None, ckpt_dir=ckpt_dir, expname="exp", no_reload=True,
                     perturb=1.0, white_bkgd=True, raw_noise_std=0.0, mode="linear", color_mode="midpoint",
                     dataset="blender", no_ndc=False, lindisp=False, precision=precision, lrate_decay=lrate_decay,
                     constant_init=0, chunk=32768, N_rand=n_rand)


def train_psnr(P, scene, precision, steps, dev, rays=4096, seed=0, tail_frac=0.1, init=None):
    """Train both networks for `steps` steps on the analytic scene in `precision` (weights from `init`: two state
    dicts, or torch.manual_seed(0)'s default initialisation; pixel choice, jitter and sampler draws are counter-based
    functions of (seed, step, ray): identical in every precision).  Returns the training PSNR of the fine image over
    the last `tail_frac` of the steps (mean of the per-step values run_plnerf.py:1290 prints), the held-out view's
    PSNR after the last step, the loss curve's marks and the time per step."""
    ck = tempfile.mkdtemp()
    os.makedirs(os.path.join(ck, "exp"))
    args = nerf_args(precision, ck, n_rand=rays)
    torch.manual_seed(0)
    so, sys.stdout = sys.stdout, open(os.devnull, "w")
    try:
        kw, kw_test, _, _, opt, opt_c = P.create_nerf(args, device=dev)
    finally:
        sys.stdout = so
    if init is not None:
        kw["network_fn"].load_state_dict(init[0])
        kw["network_fine"].load_state_dict(init[1])
    ts = P.TrainStep(args, kw, opt, opt_c, distributed=False, seed=seed)
    psnrs, losses = [], []
    torch.cuda.synchronize()
    t0 = torch.cuda.Event(enable_timing=True)
    t1 = torch.cuda.Event(enable_timing=True)
    t0.record()
    for i in range(steps):
        v = i % len(scene.poses)
        loss, psnr = ts.step_view(scene.H, scene.W, scene.K, scene.poses[v], scene.images[v], near=scene.near,
                                  far=scene.far, n_rand=rays)
        psnrs.append(psnr)
        losses.append(loss)
    t1.record()
    torch.cuda.synchronize()
    ms = t0.elapsed_time(t1) / steps
    psnrs = torch.stack(psnrs).double().cpu()
    losses = torch.stack(losses).double().cpu()
    tail = max(1, int(round(steps * tail_frac)))
    with torch.no_grad():
        kw_eval = dict(kw_test, perturb=0.0)      # deterministic evaluation render
        rgb, _, _, _ = P.render(scene.h_hw, scene.h_hw, scene.h_K, chunk=65536, c2w=scene.h_pose[:3, :4], near=scene.near,
                                far=scene.far, **kw_eval)
        mse = torch.mean((rgb - scene.h_image) ** 2)
        held = float(-10.0 * torch.log10(mse))
    ts.check_range()
    marks = [m for m in (1, 10, 50, 100, 200, 500, 1000, 2000, 5000) if m <= steps]
    return {"precision": precision, "steps": steps, "rays_per_step": rays, "ms_per_step": ms,
            "psnr_train_tail_mean": float(psnrs[-tail:].mean()), "tail_steps": tail,
            "psnr_heldout_view": held, "loss_at": {str(m): float(losses[m - 1]) for m in marks},
            "loss_tail_mean": float(losses[-tail:].mean())}


def psnr_vs_ref(P, dev, steps, rays=4096, precision="f16x3", views=8, seed=0, with_twin=False, only_run=False):
    """The benchmarked arithmetic against the exact-fp32 kernels (reference-equal gradients, pinned to the oracle at
    1e-5 by tests/test_gpu_fullsize.py) on the same scene, weights and draws: {run, ref, gap_db ...}."""
    scene = AnalyticScene(P, views, dev)
    # identical initial weights: default nn.Linear initialisation under one seed, handed to both runs as state dicts
    ck = tempfile.mkdtemp()
    os.makedirs(os.path.join(ck, "exp"))
    torch.manual_seed(0)
    so, sys.stdout = sys.stdout, open(os.devnull, "w")
    try:
        kw0, _, _, _, _, _ = P.create_nerf(nerf_args("fp32", ck, n_rand=rays), device=dev)
    finally:
        sys.stdout = so
    init = ({k: v.clone() for k, v in kw0["network_fn"].state_dict().items()},
            {k: v.clone() for k, v in kw0["network_fine"].state_dict().items()})
    del kw0
    run = train_psnr(P, scene, precision, steps, dev, rays, seed, init=init)
    if only_run:      # (a kernel variant's run alone: the fp32 reference of the same seed is already on file)
        return {"run": run}
    ref = train_psnr(P, scene, "fp32", steps, dev, rays, seed, init=init)
    twin = None
    if with_twin:
        # the noise floor of the comparison: the SAME exact-fp32 arithmetic from initial weights moved by one part in
        # 10^7 (a fraction of an fp32 ulp per weight on average, a few ulps for some) -- an optimisation of a ReLU network
        # through a discontinuous sampler is chaotic, so two runs that differ at rounding level end a few tenths of a dB
        # apart; a precision mode's gap means something only against this
        gen = torch.Generator().manual_seed(1234)
        moved = tuple({k: (v * (1.0 + 1e-7 * torch.randn(v.shape, generator=gen).to(v.device))) for k, v in sd.items()}
                      for sd in init)
        twin = train_psnr(P, scene, "fp32", steps, dev, rays, seed, init=moved)
    return {"scene": f"analytic sphere, {views} training views 800x800 on the NeRF-synthetic camera ring, white "
                     f"background; held-out view 200x200 between two training poses",
            "what": "mean training PSNR of the fine image over the last 10 % of the steps (the value run_plnerf.py:1290 "
                    "prints per step) and the held-out view's PSNR after the last step; both runs from identical weights, "
                    "pixels and draws",
            "run": run, "ref": ref,
            "gap_db_train": run["psnr_train_tail_mean"] - ref["psnr_train_tail_mean"],
            "gap_db_heldout": run["psnr_heldout_view"] - ref["psnr_heldout_view"],
            **({"ref_twin": twin,
                "noise_floor_db_train": twin["psnr_train_tail_mean"] - ref["psnr_train_tail_mean"],
                "noise_floor_db_heldout": twin["psnr_heldout_view"] - ref["psnr_heldout_view"],
                "noise_floor_what": "the same fp32 run from initial weights perturbed by 1e-7 relative"} if twin else {})}
