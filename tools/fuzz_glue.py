"""The step's glue kernels against the torch expressions they replace, BIT FOR BIT, over random shapes and magnitudes:

  * plnerf_coarse_samples (run_plnerf.py:683-708: depths linear in depth or in disparity, stratified jitter, positions),
  * plnerf_ray_points (:708, :735), plnerf_ndc_rays (run_nerf_helpers.py:184-201), plnerf_merge_sort (:731-734),
  * plnerf_image_loss (:1287-1300: both MSE terms, the psnr and both image gradients; 2e-6 relative -- a reduction),
  * Embedder via plnerf_embed_rows against the reference's encoder expression (1e-6: sin / cos of fl(fl(x s) 2^k)).

Every comparison is against torch on the HOST (the oracle's arithmetic).  python tools/fuzz_glue.py --cases 300 > out.json"""
import argparse, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import plnerf_amd as P
from plnerf_amd import functional as Fn
from plnerf_amd import rays as RAYS

ap = argparse.ArgumentParser()
ap.add_argument("--cases", type=int, default=300)
ap.add_argument("--seed", type=int, default=13)
a = ap.parse_args()
dev = torch.device("cuda:0")
rng = np.random.default_rng(a.seed)
g = lambda x: x.to(dev)
bits = lambda x, y: torch.equal(x.detach().cpu().contiguous().view(torch.int32), y.detach().contiguous().view(torch.int32))
stats = {"cases": 0, "coarse_samples": 0, "ray_points": 0, "ndc_rays": 0, "image_loss_worst": 0.0, "embed_worst": 0.0}
violations = []
for case in range(a.cases):
    R = int(rng.choice([1, 2, 33, 1000, 4096]))
    S = int(rng.choice([2, 3, 17, 64, 128, 192, 1000]))
    gen = torch.Generator().manual_seed(11000 + case)
    scale = float(rng.choice([1e-3, 1.0, 1.0, 50.0]))
    o = torch.randn(R, 3, generator=gen) * scale
    d = torch.randn(R, 3, generator=gen) * float(rng.choice([0.1, 1.0, 10.0]))
    near = torch.rand(R, 1, generator=gen) * 2.0 + 0.05
    far = near + torch.rand(R, 1, generator=gen) * 6.0 + 0.1
    t_vals = torch.linspace(0.0, 1.0, steps=S)
    lindisp, perturb = bool(rng.integers(2)), bool(rng.integers(2))
    t_rand = torch.rand(R, S, generator=gen) if perturb else None
    bad = []
    # reference expressions (run_plnerf.py:683-708)
    z = near * (1.0 - t_vals) + far * t_vals if not lindisp else 1.0 / (1.0 / near * (1.0 - t_vals) + 1.0 / far * t_vals)
    z = z.expand(R, S)
    if perturb:
        mids = 0.5 * (z[..., 1:] + z[..., :-1])
        upper, lower = torch.cat([mids, z[..., -1:]], -1), torch.cat([z[..., :1], mids], -1)
        z = lower + (upper - lower) * t_rand
    pts = o[..., None, :] + d[..., None, :] * z[..., :, None]
    z_h, pts_h = Fn.coarse_samples(g(o), g(d), g(near), g(far), g(t_vals), None if t_rand is None else g(t_rand), lindisp, perturb, None)
    stats["coarse_samples"] += z.numel()
    if not (bits(z_h, z.contiguous()) and bits(pts_h, pts.contiguous())):
        bad.append(f"coarse_samples: {int((z_h.cpu() != z).sum())} depths / {int((pts_h.cpu() != pts).sum())} coordinates differ")
    z2, _ = torch.sort(near + (far - near) * torch.rand(R, S, generator=gen), -1)
    pts2 = o[..., None, :] + d[..., None, :] * z2[..., :, None]
    stats["ray_points"] += pts2.numel()
    if not bits(Fn.ray_points(g(o), g(d), g(z2)), pts2.contiguous()):
        bad.append("ray_points differs")
    # NDC warp
    H, W_, focal, nr = int(rng.integers(8, 1200)), int(rng.integers(8, 1200)), float(rng.uniform(50.0, 1500.0)), 1.0
    dn = d.clone(); dn[:, 2] = -dn[:, 2].abs() - 0.05
    t = -(nr + o[..., 2]) / dn[..., 2]
    oo = o + t[..., None] * dn
    sx, sy = -1.0 / (W_ / (2.0 * focal)), -1.0 / (H / (2.0 * focal))
    o_ref = torch.stack([sx * oo[..., 0] / oo[..., 2], sy * oo[..., 1] / oo[..., 2], 1.0 + 2.0 * nr / oo[..., 2]], -1)
    d_ref = torch.stack([sx * (dn[..., 0] / dn[..., 2] - oo[..., 0] / oo[..., 2]), sy * (dn[..., 1] / dn[..., 2] - oo[..., 1] / oo[..., 2]),
                         -2.0 * nr / oo[..., 2]], -1)
    o_h, d_h = RAYS.ndc_rays(H, W_, focal, nr, g(o), g(dn))
    stats["ndc_rays"] += o_ref.numel()
    if not (bits(o_h, o_ref.contiguous()) and bits(d_h, d_ref.contiguous())):
        bad.append(f"ndc_rays: {int((o_h.cpu() != o_ref).sum())} / {int((d_h.cpu() != d_ref).sum())} values differ")
    # image loss + gradients
    rgb, rgb0, tgt = torch.rand(R, 3, generator=gen), torch.rand(R, 3, generator=gen), torch.rand(R, 3, generator=gen)
    l1, l0 = torch.mean((rgb - tgt) ** 2), torch.mean((rgb0 - tgt) ** 2)
    ref4 = torch.stack([l1 + l0, l1, l0, -10.0 * torch.log10(l1)])
    loss4, g1, g0 = Fn.image_loss_and_grads(g(rgb), g(rgb0), g(tgt))
    e = float(((loss4.cpu() - ref4).abs() / (1e-12 + ref4.abs())).max())
    e = max(e, float((g1.cpu() - 2.0 * (rgb - tgt) / (3 * R)).abs().max()) * 3 * R, float((g0.cpu() - 2.0 * (rgb0 - tgt) / (3 * R)).abs().max()) * 3 * R)
    stats["image_loss_worst"] = max(stats["image_loss_worst"], e)
    if e > 2e-6:
        bad.append(f"image_loss {e:.2e}")
    # encoder rows
    if R * S <= 200000:
        fx, fd = int(rng.integers(0, 11)), int(rng.integers(0, 5))
        sc = float(rng.choice([1.0, np.pi]))
        p3 = (torch.rand(R, S, 3, generator=gen) * 2 - 1) * 1.5
        v3 = torch.nn.functional.normalize(torch.randn(R, 3, generator=gen), dim=-1)
        def enc(x, L):
            outs = [x]
            for k in range(L):
                arg = (x * np.float32(sc)) * np.float32(2.0 ** k)
                outs += [torch.sin(arg), torch.cos(arg)]
            return torch.cat(outs, -1)
        ref_e = torch.cat([enc(p3.reshape(-1, 3), fx), enc(v3[:, None].expand(R, S, 3).reshape(-1, 3), fd)], -1)
        got_e = Fn.embed_rows(g(p3), g(v3), None, fx, fd, input_scale=sc).cpu()
        ee = float((got_e - ref_e).abs().max())
        stats["embed_worst"] = max(stats["embed_worst"], ee)
        if got_e.shape != ref_e.shape or ee > 1e-6:
            bad.append(f"embed_rows {ee:.2e}")
    stats["cases"] += 1
    if bad:
        violations.append({"case": case, "R": R, "S": S, "lindisp": lindisp, "perturb": perturb, "what": bad})
print(json.dumps({"what": "glue kernels vs the torch expressions they replace", "seed": a.seed, "stats": stats, "violations": violations}))
sys.exit(1 if violations else 0)
