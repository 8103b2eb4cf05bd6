"""Quadrature campaign: plnerf_quad_fwd / plnerf_quad_bwd / plnerf_quad_bwd_rays (raw2outputs, run_plnerf.py:553-624, with
compute_weights_piecewise_linear :516-550 and compute_weights :504-513) against the CPU oracle over random shapes (1-2048
rays, 2-1022 samples = PLNERF_MAX_SAMPLES), both rules and colour rules, white background, farcolorfix, density noise, and adversarial densities
(empty space, opaque slabs, densities of 1e4, zero-length and unit-length intervals, |d| from 1e-2 to 1e2).

Bounds: every forward output (rgb, disp, acc, depth, weights, tau, T) 1e-5 abs + rel against the oracle in fp32, NaN patterns
equal; d / d raw and d / d (z_vals, near, far, rays_d) against the oracle's fp64 autograd within four times the fp32 oracle's
own distance from it (on an almost empty ray disp = 1 / max(1e-10, depth / acc) hands back (1 - z acc / depth) / depth, which
cancels in ANY fp32 evaluation: seed 2026 has a case at 2.8e-3 against the oracle's own 1.0e-3) + 2e-5 of max |g| up to 256 samples (BASELINE's largest count is 192), + 1e-3 beyond (the prefix products and
the reverse scan run in fp32 over up to 1022 elements; measured worst 6e-4 at 1022).
Test infrastructure (imports oracle/).  python tools/fuzz_quadrature.py --cases 300 --seed 9 > out.json"""
import argparse, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import plnerf_amd as P
from plnerf_amd.functional import QuadratureFn
from oracle import plnerf_oracle as orc

ap = argparse.ArgumentParser()
ap.add_argument("--cases", type=int, default=300)
ap.add_argument("--seed", type=int, default=9)
a = ap.parse_args()
torch.set_num_threads(min(16, os.cpu_count() or 1))
dev = torch.device("cuda:0")
rng = np.random.default_rng(a.seed)
g = lambda x: x.to(dev)
FWD_TOL = 1e-5
NAMES = ("rgb", "disp", "acc", "weights", "depth", "tau", "T")
stats = {"cases": 0, "rays": 0, "forward_worst": 0.0, "g_raw_worst": 0.0, "g_raw_oracle32": 0.0, "geometry_worst": 0.0, "geometry_oracle32": 0.0}
violations = []
for case in range(a.cases):
    R = int(rng.choice([1, 3, 67, 512, 2048]))
    S = int(rng.choice([2, 3, 5, 37, 64, 65, 128, 192, 257, 700, 1022]))
    if R * S > 300000:
        R = max(1, 300000 // S)
    mode = ["linear", "linear", "constant"][int(rng.integers(3))]
    cm = ["midpoint", "left"][int(rng.integers(2))] if mode == "linear" else "midpoint"
    wb, fcf, noisy = bool(rng.integers(2)), bool(rng.integers(2)), bool(rng.integers(2))
    gen = torch.Generator().manual_seed(9000 + case)
    raw = torch.randn(R, S, 4, generator=gen)
    kind = int(rng.integers(5))
    raw[..., 3] = raw[..., 3] * 4.0 + (1.0 if kind != 1 else -8.0)
    if kind == 2 and S >= 8:
        raw[: max(R // 2, 1), S // 4: S // 2, 3] += 30.0
    if kind == 3:
        raw[..., 3] = raw[..., 3].abs() * 1e3                                  # huge densities
    z, _ = torch.sort(2.0 + 4.0 * torch.rand(R, S, generator=gen), -1)
    if kind == 4 and S >= 4:
        z[:, 1] = z[:, 0]                                                      # a zero-length interval
    near = torch.full((R, 1), 2.0) - 0.3 * torch.rand(R, 1, generator=gen)
    far = torch.full((R, 1), 6.0) + 0.3 * torch.rand(R, 1, generator=gen)
    d = torch.randn(R, 3, generator=gen) * float(rng.choice([1e-2, 1.0, 1.0, 1e2]))
    noise = torch.randn(R, S, generator=gen) if noisy else None
    n = S + 1 if mode == "linear" else S
    cot = [torch.randn(R, 3, generator=gen), torch.randn(R, generator=gen) * 0.1, torch.randn(R, generator=gen),
           torch.randn(R, n, generator=gen) * 0.1, torch.randn(R, generator=gen)]
    bad = []

    def oracle(dt):
        leaves = [t.to(dt).clone().requires_grad_(True) for t in (raw, z, near, far, d)]
        out = orc.raw2outputs(*leaves, mode, cm, white_bkgd=wb, farcolorfix=fcf, noise=None if noise is None else noise.to(dt))
        loss = sum((o * c.to(dt)).sum() for o, c in zip(out[:5], cot) if bool(torch.isfinite(o).all()))
        loss.backward()
        return out, [l.grad.double() if l.grad is not None else None for l in leaves]
    out32, g32 = oracle(torch.float32)
    _, g64 = oracle(torch.float64)
    leaves_h = [g(t).clone().requires_grad_(True) for t in (raw, z, near, far, d)]
    out_h = QuadratureFn.apply(*leaves_h, None if noise is None else g(noise), mode, cm, wb, fcf)
    e_f = 0.0
    for name, x, y in zip(NAMES, out_h, out32):
        if y is None:
            continue
        x, y = x.detach().cpu(), y.detach()
        if not torch.equal(torch.isnan(x), torch.isnan(y)):
            rows = (torch.isnan(x) != torch.isnan(y)).reshape(x.shape[0], -1).any(-1).nonzero().flatten().tolist()
            acc_h, acc_o = out_h[2].detach().cpu(), out32[2].detach()
            detail = [(r_, float(acc_h[r_]), float(acc_o[r_])) for r_ in rows[:4]]
            # disp = 1 / max(1e-10, depth / acc) is 0 / 0 = NaN on a ray whose every alpha is exactly 0, and finite as soon as ONE alpha
            # is 1 ulp: 1 - exp(-x) at x ~ 3e-8 sits on the rounding midpoint of exp, where two expf implementations inside 1 ulp
            # disagree (fourth seeds, case 172: one ray of 257,128).  Such rays -- acc below 1e-6 on both sides -- are not a pattern
            # difference of the kernels.
            if name == "disp" and all(max(abs(a_), abs(b_)) < 1e-6 for _, a_, b_ in [(r_, float(acc_h[r_]), float(acc_o[r_])) for r_ in rows]):
                stats["disp_nan_on_rays_with_acc_below_1e-6"] = stats.get("disp_nan_on_rays_with_acc_below_1e-6", 0) + len(rows)
                x, y = torch.nan_to_num(x, nan=0.0), torch.nan_to_num(y, nan=0.0)
            else:
                bad.append(f"{name}: NaN pattern differs on rays {rows[:4]} (acc path / oracle: {detail})")
                continue
        err = torch.nan_to_num((x.double() - y.double()).abs() / (1.0 + y.double().abs()), nan=0.0, posinf=0.0)
        if name == "disp":      # 1 / max(1e-10, depth / acc): on an (almost) empty ray the quotient of two roundings -- judged where
            err = err[out32[2].detach() > 1e-2]      # the ray holds something (the tests bound it by its propagated error)
        e = float(err.max()) if err.numel() else 0.0
        e_f = max(e_f, e)
        if e > FWD_TOL * max(1.0, S / 512.0):      # (1e-5 up to 512 samples -- BASELINE's largest count is 192 -- 2e-5 at 1022)
            bad.append(f"{name}: {e:.2e}")
    stats["forward_worst"] = max(stats["forward_worst"], e_f)
    loss_h = sum((o * g(c)).sum() for o, c, o32 in zip(out_h[:5], cot, out32[:5]) if bool(torch.isfinite(o32).all()))
    if isinstance(loss_h, torch.Tensor) and loss_h.requires_grad:
        loss_h.backward()
        for name, lh, r64, r32 in zip(("raw", "z_vals", "near", "far", "rays_d"), leaves_h, g64, g32):
            if r64 is None or (mode == "constant" and name in ("near", "far")):
                continue
            if not (bool(torch.isfinite(r64).all()) and bool(torch.isfinite(r32).all())):
                continue      # (the reference's own gradient overflows: densities of 1e4 over unit lengths)
            scale = float(r64.abs().max())
            if scale < 1e-12:      # (nothing, or next to nothing, flows: e.g. the far bound behind an opaque slab)
                continue
            e_h = float((lh.grad.cpu().double() - r64).abs().max()) / scale
            e_o = float((r32 - r64).abs().max()) / scale
            key = "g_raw" if name == "raw" else "geometry"
            stats[key + "_worst"], stats[key + "_oracle32"] = max(stats[key + "_worst"], e_h), max(stats[key + "_oracle32"], e_o)
            if not bool(torch.isfinite(lh.grad).all()) or e_h > 4 * e_o + (2e-5 if S <= 256 else 1e-3):
                bad.append(f"d/d {name}: HIP {e_h:.2e} vs fp32 oracle {e_o:.2e} (of max |g| {scale:.3g})")
    stats["cases"] += 1
    stats["rays"] += R
    if bad:
        violations.append({"case": case, "R": R, "S": S, "mode": mode, "color_mode": cm, "white_bkgd": wb, "farcolorfix": fcf,
                           "noise": noisy, "kind": kind, "what": bad})
print(json.dumps({"what": "quadrature campaign vs the CPU oracle", "seed": a.seed, "stats": stats, "violations": violations}))
sys.exit(1 if violations else 0)
