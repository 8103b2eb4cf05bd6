"""Sampler / sort campaign on identical inputs: plnerf_sample_const (sample_pdf, run_nerf_helpers.py:241-284), plnerf_sample_pl
(sample_pdf_reformulation, :364-445), plnerf_merge_sort (run_plnerf.py:731-734) and the fused plnerf_coarse_epilogue against the
CPU oracle over random shapes (rays 1-4096, S 2-256, N 1-256) and adversarial weights (empty stretches, opaque slabs, exact
zeros, constant densities, jitter-free and random draws incl. u = 0).

Bounds: search indices BIT-EXACT (torch.searchsorted on the same cdf); sample_pdf values 1e-4; sample_pdf_reformulation values
1e-5 on 99.99 % and 1e-3 on all (the closed form cancels on a few draws in a million: DESIGN.md section 6); merged depths equal
to torch.sort(cat(z, clamp(samples))) as VALUES bit for bit; the fused epilogue equal to the separate launches bit for bit.
Test infrastructure (imports oracle/).  python tools/fuzz_samplers.py --cases 200 --seed 3 > out.json"""
import argparse, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import plnerf_amd as P
from plnerf_amd import functional as Fn
from oracle import plnerf_oracle as orc

ap = argparse.ArgumentParser()
ap.add_argument("--cases", type=int, default=200)
ap.add_argument("--seed", type=int, default=3)
a = ap.parse_args()
torch.set_num_threads(min(16, os.cpu_count() or 1))
dev = torch.device("cuda:0")
rng = np.random.default_rng(a.seed)
g = lambda x: x.to(dev)
stats = {"cases": 0, "rays": 0, "const_indices": 0, "const_worst": 0.0, "pl_indices": 0, "pl_beyond_1e-5": 0, "pl_worst": 0.0,
         "sort_values": 0, "fused_epilogue_runs": 0}
violations = []
for case in range(a.cases):
    R = int(rng.choice([1, 2, 5, 64, 333, 1024, 4096]))
    S = int(rng.integers(2, 257))
    N = int(rng.integers(1, 257))
    gen = torch.Generator().manual_seed(7000 + case)
    raw = torch.randn(R, S, 4, generator=gen)
    kind = int(rng.integers(5))
    raw[..., 3] = raw[..., 3] * 4.0 + (1.0 if kind != 1 else -6.0)          # (kind 1: almost empty space)
    if kind == 2 and S >= 8:
        raw[: max(R // 2, 1), S // 4: S // 2, 3] += 30.0                     # opaque slab
    if kind == 3:
        raw[..., 3] = torch.round(raw[..., 3])                               # repeated / exactly equal densities
    if kind == 4:
        raw[..., 3] = 0.5                                                    # constant density: the flat-interval branch everywhere
    z, _ = torch.sort(2.0 + 4.0 * torch.rand(R, S, generator=gen), -1)
    near, far = torch.full((R, 1), 2.0), torch.full((R, 1), 6.0)
    d = torch.randn(R, 3, generator=gen)
    det = bool(rng.integers(3) == 0)
    u = torch.linspace(0.0, 1.0, N) if det else torch.rand(R, N, generator=gen)
    if not det and R * N > 4:
        u.view(-1)[:2] = torch.tensor([0.0, 0.999999])                      # the ends of the cdf
    u_o = u.expand(R, N).contiguous() if det else u      # (the oracle takes one row per ray; the kernels also a shared row)
    white = bool(rng.integers(2))
    cm = ["midpoint", "left"][int(rng.integers(2))]
    with torch.no_grad():
        _, _, _, w, _, tau, T = orc.raw2outputs(raw, z, near, far, d, "linear", cm, white_bkgd=white)
        wc = orc.raw2outputs(raw, z, near, far, d, "constant", "midpoint", white_bkgd=white)[3]
        bad = []
        # sample_pdf on identical bins / weights
        if S >= 4:
            bins = 0.5 * (z[..., 1:] + z[..., :-1])
            s_o, i_o = orc.sample_pdf(bins, wc[..., 1:-1], N, u=u_o, return_inds=True)
            s_h, i_h = Fn.sample_const(g(bins), g(wc[..., 1:-1]), g(u), want_inds=True)
            stats["const_indices"] += i_o.numel()
            if not torch.equal(i_o, i_h.cpu()):
                bad.append(f"sample_pdf: {int((i_o != i_h.cpu()).sum())} indices differ")
            e = float((s_h.cpu() - s_o).abs().max())
            stats["const_worst"] = max(stats["const_worst"], e)
            if e > 1e-4:      # (t = (u - cdf) / (cdf step): an ulp of the cdf over a step of 1e-3 ... the indices are the contract)
                bad.append(f"sample_pdf values {e:.2e}")
        # sample_pdf_reformulation on identical weights / tau / T
        s_o, _, _, _, i_o = orc.sample_pdf_reformulation(z, w, tau, T, near, far, N, u=u_o, return_inds=True)
        s_h, i_h = Fn.sample_pl(g(z), g(w), g(tau), g(T), g(near), g(far), g(u), 1e-4, 1e-3, want_inds=True)
        stats["pl_indices"] += i_o.numel()
        if not torch.equal(i_o, i_h.cpu()):
            bad.append(f"sample_pdf_reformulation: {int((i_o != i_h.cpu()).sum())} indices differ")
        dd = (s_h.cpu().double() - s_o.double()).abs()
        nanpat = torch.equal(torch.isnan(s_h.cpu()), torch.isnan(s_o))
        dd = torch.nan_to_num(dd, nan=0.0)
        stats["pl_beyond_1e-5"] += int((dd > 1e-5 * (1.0 + s_o.double().abs().nan_to_num())).sum())
        stats["pl_worst"] = max(stats["pl_worst"], float(dd.max()))
        if float(dd.max()) > 1e-3 or not nanpat:
            bad.append(f"sample_pdf_reformulation values {float(dd.max()):.2e} (NaN pattern equal: {nanpat})")
        # clamp + cat + sort
        if S + N <= 1024:
            ref_sorted = torch.sort(torch.cat([z, torch.clamp(s_o, near, far)], -1), -1)[0]
            got_sorted = Fn.merge_sort(g(z), g(s_o), g(near), g(far)).cpu()
            stats["sort_values"] += ref_sorted.numel()
            if not torch.equal(ref_sorted.view(torch.int32), got_sorted.view(torch.int32)):
                bad.append("merge_sort differs from torch.sort(cat(clamp))")
        # the fused coarse epilogue against the separate launches
        if S + N <= 1024:
            o3 = torch.randn(R, 3, generator=gen)
            fused = Fn.CoarseEpilogueFn.apply(g(raw), g(z), g(near), g(far), g(o3), g(d), None, g(u), N, cm, white, False, 1e-4, 1e-3,
                                              None, False)
            rgb, disp, acc, wq, depth, tq, Tq = Fn.QuadratureFn.apply(g(raw), g(z), g(near), g(far), g(d), None, "linear", cm, white, False)
            sq = Fn.sample_pl(g(z), wq, tq, Tq, g(near), g(far), g(u), 1e-4, 1e-3)
            zq = Fn.merge_sort(g(z), sq, g(near), g(far))
            same = all(torch.equal(x.view(torch.int32), y.view(torch.int32)) for x, y in ((fused[0], rgb), (fused[2], acc), (fused[3], depth), (fused[4], zq)))
            stats["fused_epilogue_runs"] += 1
            if not same:
                bad.append("fused coarse epilogue differs from the separate launches")
    stats["cases"] += 1
    stats["rays"] += R
    if bad:
        violations.append({"case": case, "R": R, "S": S, "N": N, "kind": kind, "det": det, "what": bad})
if stats["pl_indices"] and stats["pl_beyond_1e-5"] > 1e-4 * stats["pl_indices"]:
    violations.append({"what": f"{stats['pl_beyond_1e-5']} of {stats['pl_indices']} reformulation samples beyond 1e-5"})
print(json.dumps({"what": "sampler / sort campaign on identical inputs vs the CPU oracle", "seed": a.seed, "stats": stats, "violations": violations}))
sys.exit(1 if violations else 0)
