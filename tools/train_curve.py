"""Optimisation sanity ("PSNR vs reference" proxy): train the two networks for K steps on an
analytic scene (rays of a fixed camera ring, target colour = a smooth function of the ray) in the exact fp32 mode
and in a 16-bit mode, from identical initial weights and identical random draws, and print both loss curves.
    python tools/train_curve.py [--steps 300] [--modes fp32,f16x3,bf16]"""
import argparse, json, os, sys, tempfile
from argparse import Namespace
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import plnerf_amd as P

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=300)
ap.add_argument("--rays", type=int, default=2048)
ap.add_argument("--modes", default="fp32,f16x3,bf16")
a = ap.parse_args()
dev = torch.device("cuda:0")


def batch_for(step):
    """Rays of view (step mod 40) and an analytic target: a soft sphere of radius 1 with a colour gradient."""
    batch, _, K = P.rays.synthetic_blender_rays(a.rays, seed=step, theta=-180.0 + 9.0 * (step % 40), device="cpu")
    o, d = batch[0], batch[1]
    dn = d / d.norm(dim=-1, keepdim=True)
    b = (o * dn).sum(-1)
    disc = b * b - ((o * o).sum(-1) - 1.0)
    hit = disc > 0
    t = -b - torch.sqrt(torch.clamp(disc, min=0))
    p = o + dn * t[:, None]
    col = torch.where(hit[:, None], 0.5 + 0.5 * p, torch.ones_like(p))       # white background
    return (o.to(dev), d.to(dev)), col.to(dev), K


curves = {}
for mode in a.modes.split(","):
    ck = tempfile.mkdtemp(); os.makedirs(os.path.join(ck, "exp"))
    args = Namespace(multires=10, i_embed=0, use_viewdirs=True, multires_views=4, N_importance=128, N_samples=64,
                     netdepth=8, netwidth=256, netdepth_fine=8, netwidth_fine=256, netchunk=1 << 22, lrate=5e-4,
                     coarse_lrate=5e-4, ft_path=None, ckpt_dir=ck, expname="exp", no_reload=True, perturb=1.0,
                     white_bkgd=True, raw_noise_std=0.0, mode="linear", color_mode="midpoint", dataset="blender",
                     no_ndc=False, lindisp=False, lrate_decay=500, constant_init=0, chunk=32768, precision=mode)
    torch.manual_seed(0)
    sys.stdout, so = open(os.devnull, "w"), sys.stdout
    kw, _, _, _, opt, opt_c = P.create_nerf(args, device=dev)
    sys.stdout = so
    step = P.TrainStep(args, kw, opt, opt_c, distributed=False)
    torch.manual_seed(1)                       # the stratified / importance draws
    losses = []
    for i in range(a.steps):
        rays, target, K = batch_for(i)
        loss, psnr = step(800, 800, K, rays, target, near=2.0, far=6.0)
        losses.append(float(loss))
    curves[mode] = losses
    marks = [0, 9, 49, 99, 199, a.steps - 1]
    print(json.dumps({"mode": mode, "steps": a.steps, "rays": a.rays,
                      "loss_at": {str(m + 1): round(losses[m], 6) for m in marks if m < a.steps},
                      "mean_last_20": round(float(np.mean(losses[-20:])), 6)}), flush=True)
ref = curves.get("fp32")
if ref is not None:
    for mode, c in curves.items():
        if mode == "fp32":
            continue
        rel = [abs(x - y) / max(y, 1e-12) for x, y in zip(c, ref)]
        print(json.dumps({"mode": mode, "vs": "fp32", "max_rel_loss_diff_first_50": round(max(rel[:50]), 5),
                          "max_rel_loss_diff_all": round(max(rel), 5),
                          "rel_diff_mean_last_20": round(abs(np.mean(c[-20:]) - np.mean(ref[-20:])) / np.mean(ref[-20:]), 5)}))
