"""Full-image inference: render(H, W, K, chunk, c2w=pose) of one 800x800
view -- 640,000 rays in 32,768-ray chunks, coarse 64 + fine 192 samples, no_grad -- per precision."""
import argparse, json, os, sys, tempfile
from argparse import Namespace
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import plnerf_amd as P

ap = argparse.ArgumentParser()
ap.add_argument("--precisions", default="fp32,f16x3,f16")
ap.add_argument("--frames", type=int, default=2)
ap.add_argument("--chunk", type=int, default=32768)
a = ap.parse_args()
dev = torch.device("cuda:0")
H = W = 800
focal = 0.5 * W / torch.tan(torch.tensor(0.5 * 0.6911112)).item()
K = [[focal, 0, 0.5 * W], [0, focal, 0.5 * H], [0, 0, 1]]
for prec in a.precisions.split(","):
    d = tempfile.mkdtemp(); os.makedirs(os.path.join(d, "exp"))
    args = Namespace(multires=10, i_embed=0, use_viewdirs=True, multires_views=4, N_importance=128, N_samples=64,
                     netdepth=8, netwidth=256, netdepth_fine=8, netwidth_fine=256, netchunk=65536, lrate=5e-4,
                     coarse_lrate=5e-4, ft_path=None, ckpt_dir=d, expname="exp", no_reload=True, perturb=1.0,
                     white_bkgd=True, raw_noise_std=0.0, mode="linear", color_mode="midpoint", dataset="blender",
                     no_ndc=False, lindisp=False, precision=prec)
    so = sys.stdout; sys.stdout = open(os.devnull, "w")
    torch.manual_seed(0)
    _, kw_test, _, _, _, _ = P.create_nerf(args, device=dev)
    sys.stdout = so
    poses = [P.rays.pose_spherical(th, -30.0, 4.0)[:3, :4].to(dev) for th in (0.0, 90.0, 180.0)]
    with torch.no_grad():
        P.render(H, W, K, chunk=a.chunk, c2w=poses[0], near=2.0, far=6.0, **kw_test)   # warm-up
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for f in range(a.frames):
            rgb, disp, acc, _ = P.render(H, W, K, chunk=a.chunk, c2w=poses[(f + 1) % 3], near=2.0, far=6.0, **kw_test)
        e.record(); torch.cuda.synchronize()
    ms = s.elapsed_time(e) / a.frames
    # the frame's algorithmic MLP work (1,186,816 FLOP per network evaluation, 64 + 192 per ray)
    # against the dense MFMA peak of the operand type: the frame IS the two MLP launches per chunk (everything else --
    # ray generation, sampling, the fused coarse epilogue, quadrature -- is < 1 % of it)
    tflops = H * W * (64 + 192) * 1186816 / (ms * 1e-3) / 1e12
    peak = 157.3 if prec == "fp32" else 2500.0
    print(json.dumps({"what": "render 800x800 frame (64+192 samples, inference)", "precision": prec,
                      "ms_per_frame": ms, "rays_per_s": H * W / (ms * 1e-3), "chunk": a.chunk,
                      "roofline": {"bound": "mfma", "achieved": tflops, "peak": peak, "unit": "TFLOP/s",
                                   "frac": tflops / peak, "note": "whole frame: algorithmic MLP FLOP / frame time"},
                      "finite": bool(torch.isfinite(rgb).all())}), flush=True)
