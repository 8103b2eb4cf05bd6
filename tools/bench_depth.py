"""Depth-supervised variant (BASELINE config 5: mode=linear, N_samples=128 / N_importance=64, space-carving loss
through pred_hyp): rays/s of the full training step on one GPU, synthetic Blender-style rays and hypotheses.
    python tools/bench_depth.py [--rays 4096] [--precision f16x3]"""
import argparse, json, os, sys, time
from argparse import Namespace
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import plnerf_amd as P
from plnerf_amd import depth as Dp

ap = argparse.ArgumentParser()
ap.add_argument("--rays", type=int, default=4096)
ap.add_argument("--steps", type=int, default=10)
ap.add_argument("--warmup", type=int, default=3)
ap.add_argument("--precisions", default="fp32,f16x3,bf16")
ap.add_argument("--n-samples", type=int, default=128)
ap.add_argument("--n-importance", type=int, default=64)
a = ap.parse_args()
dev = torch.device("cuda:0")
batch, target, K = P.rays.synthetic_blender_rays(a.rays, seed=0, device="cpu")
rays_o, rays_d = batch[0], batch[1]
vd = rays_d / rays_d.norm(dim=-1, keepdim=True)
ray_batch = torch.cat([rays_o, rays_d, torch.full((a.rays, 1), 2.0), torch.full((a.rays, 1), 6.0), vd], -1).to(dev)
target = target.to(dev)
gen = torch.Generator().manual_seed(0)
target_h = (2.0 + 4.0 * torch.rand(3, a.rays, 1, generator=gen)).to(dev)       # 3 depth hypotheses per ray
for prec in a.precisions.split(","):
    args = Namespace(multires=9, i_embed=0, use_viewdirs=True, multires_views=0, input_ch_cam=0,
                     N_importance=a.n_importance, N_samples=a.n_samples, netdepth=8, netwidth=256, netdepth_fine=8,
                     netwidth_fine=256, netchunk=1 << 22, lrate=5e-4, perturb=1.0, white_bkgd=True, raw_noise_std=0.0,
                     mode="linear", color_mode="midpoint", lindisp=False, no_reload=True, space_carving_weight=0.007,
                     warm_start_nerf=0, is_joint=False, norm_p=2, space_carving_threshold=0.0, precision=prec)
    torch.manual_seed(0)
    kw, _, _, grad_vars, opt = Dp.create_nerf(args, device=dev)
    step = Dp.DepthTrainStep(args, kw, opt, grad_vars, distributed=False)
    for _ in range(a.warmup):
        step(ray_batch, target, target_h)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        loss, img_loss, sc, _ = step(ray_batch, target, target_h)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.steps
    print(json.dumps({"what": "depth-supervised training step (render + pred_hyp + space carving + backward + clipped Adam)",
                      "precision": prec, "rays": a.rays, "samples": f"{a.n_samples}+{a.n_importance}",
                      "ms_per_step": 1e3 * dt, "rays_per_s": a.rays / dt, "loss": float(loss),
                      "space_carving_loss": float(sc)}), flush=True)
