"""The depth-supervised variant's render_rays (depth_supervised_exps/run_nerf_sample_based_depth.py:792-958) under the same
differential campaign as tools/fuzz_render_rays.py: the HIP path against the CPU oracle on the reference's pytest=True draws,
exact fp32 and f16x3, per stage:

  * coarse maps (rgb0 / acc0 / depth0) against the oracle's own run: 1e-5 on every ray;
  * the fine stage on IDENTICAL samples -- the returned z_vals through the oracle's fine network and quadrature: 1e-5;
  * (piecewise-linear mode) the depth hypotheses' sampler on the path's own final weights / tau / T / u: search indices
    bit-exact, values 1e-5 on 99.99 % and 1e-3 on all (the closed form cancels on a few draws in a million);
  * the one-launch stages against the separate launches (STAGE_TAP route): bit for bit.

Test infrastructure (imports oracle/).  python tools/fuzz_render_rays_depth.py --cases 150 --seed 11 > out.json"""
import argparse, json, os, sys
from argparse import Namespace
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import plnerf_amd as P
from plnerf_amd import depth as Dp
from oracle import plnerf_oracle as orc

ap = argparse.ArgumentParser()
ap.add_argument("--cases", type=int, default=150)
ap.add_argument("--seed", type=int, default=11)
ap.add_argument("--precisions", default="fp32,f16x3")
a = ap.parse_args()
dev = torch.device("cuda:0")
rng = np.random.default_rng(a.seed)
DMOD = sys.modules[Dp.__name__]
sds = [orc.closed_form_state_dict_depth(s, True) for s in (0, 1)]


def setup(precision):
    args = Namespace(multires=9, i_embed=0, use_viewdirs=True, multires_views=0, input_ch_cam=0, N_importance=32, N_samples=32,
                     netdepth=8, netwidth=256, netdepth_fine=8, netwidth_fine=256, netchunk=65536, lrate=5e-4, perturb=1.0,
                     white_bkgd=True, raw_noise_std=0.0, mode="linear", color_mode="midpoint", lindisp=False, no_reload=True,
                     space_carving_weight=0.05, warm_start_nerf=0, is_joint=False, norm_p=2, space_carving_threshold=0.0,
                     precision=precision, bb_center=0.0, bb_scale=1.0)
    so = sys.stdout; sys.stdout = open(os.devnull, "w")
    try:
        kw = Dp.create_nerf(args, device=dev)[0]
    finally:
        sys.stdout = so
    kw["network_fn"].load_state_dict(sds[0]); kw["network_fine"].load_state_dict(sds[1])
    return kw


kws = {p: setup(p) for p in a.precisions.split(",")}
worst = {p: {"coarse": 0.0, "fine_stage_on_identical_samples": 0.0, "hypotheses_beyond_1e-5": 0, "hypotheses": 0,
             "hypotheses_worst": 0.0, "rays": 0} for p in kws}
violations = []


def relerr(x, y):
    x, y = x.detach().cpu().double(), y.double()
    return float(((x - y).abs() / (1.0 + y.abs())).max()) if x.numel() else 0.0


for case in range(a.cases):
    mode = ["linear", "linear", "constant"][int(rng.integers(3))]
    cfg = dict(N_samples=int(rng.choice([8, 17, 32, 64, 128])), N_importance=int(rng.choice([4, 9, 32, 64, 128])), mode=mode,
               color_mode=["midpoint", "left"][int(rng.integers(2))] if mode == "linear" else "midpoint",
               white_bkgd=bool(rng.integers(2)), raw_noise_std=float(rng.choice([0.0, 1.0])), perturb=1.0)
    R = int(rng.choice([1, 3, 7, 33, 64, 130]))
    batch, _ = orc.synthetic_blender_rays(R, seed=3000 + case)
    near, far = batch[:, 6:7], batch[:, 7:8]
    with torch.no_grad():
        ref = orc.render_rays_depth(batch, sds[0], sds[1], cfg["N_samples"], mode, cfg["color_mode"], perturb=1.0,
                                    N_importance=cfg["N_importance"], white_bkgd=cfg["white_bkgd"],
                                    raw_noise_std=cfg["raw_noise_std"], pytest=True)
    for prec, kw in kws.items():
        call = dict(kw, **cfg)
        with torch.no_grad():
            ret = Dp.render_rays(batch.to(dev), retraw=True, pytest=True, **call)
            tap = {}
            DMOD.STAGE_TAP = tap
            try:
                ret_t = Dp.render_rays(batch.to(dev), retraw=True, pytest=True, **call)
            finally:
                DMOD.STAGE_TAP = None
        bad = []
        same = all(torch.equal(ret[k].view(torch.int32), ret_t[k].view(torch.int32))
                   for k in ("rgb_map", "acc_map", "depth_map", "pred_hyp", "z_vals", "rgb0", "weights"))
        if not same:
            bad.append("one-launch and separate-launch stages differ")
        e_c = max(relerr(ret[k], ref[k]) for k in ("rgb0", "acc0", "depth0"))
        with torch.no_grad():
            fs = orc.fine_stage(batch, sds[1], ret["z_vals"].cpu(), mode, cfg["color_mode"], cfg["white_bkgd"],
                                cfg["raw_noise_std"], True, depth_variant=True)
        e_s = max(relerr(ret[k], fs[k]) for k in ("rgb_map", "acc_map", "depth_map"))
        w = worst[prec]
        w["coarse"], w["fine_stage_on_identical_samples"] = max(w["coarse"], e_c), max(w["fine_stage_on_identical_samples"], e_s)
        w["rays"] += R
        if e_c > 1e-5:
            bad.append(f"coarse maps {e_c:.2e}")
        if e_s > 1e-5:
            bad.append(f"fine stage on identical samples {e_s:.2e}")
        if mode == "linear":
            t = {k: v.detach().cpu() for k, v in tap.items()}
            with torch.no_grad():
                s_o, _, _, _, inds_o = orc.sample_pdf_reformulation(ret["z_vals"].cpu(), t["weights_full"], t["tau"], t["T"], near, far,
                                                                    cfg["N_importance"], u=ret["u"].cpu(), return_inds=True)
            if not torch.equal(inds_o, t["hyp_inds"]):
                bad.append(f"{int((inds_o != t['hyp_inds']).sum())} hypothesis search indices differ on identical inputs")
            d = (ret["pred_hyp"].cpu().double() - s_o.double()).abs()
            n_bad = int((d > 1e-5 * (1.0 + s_o.double().abs())).sum())
            w["hypotheses_beyond_1e-5"] += n_bad
            w["hypotheses"] += d.numel()
            w["hypotheses_worst"] = max(w["hypotheses_worst"], float(d.max()))
            if float(d.max()) > 1e-3:
                bad.append(f"a hypothesis {float(d.max()):.2e} from the oracle's on identical inputs")
        if bad:
            violations.append({"case": case, "precision": prec, "R": R, "cfg": cfg, "what": bad})
for w in worst.values():
    if w["hypotheses"] and w["hypotheses_beyond_1e-5"] > 1e-4 * w["hypotheses"]:
        violations.append({"what": f"{w['hypotheses_beyond_1e-5']} of {w['hypotheses']} hypotheses beyond 1e-5 (bound: 1e-4 of them)"})
print(json.dumps({"what": "depth-supervised render_rays differential campaign vs the CPU oracle (pytest=True draws)", "cases": a.cases,
                  "seed": a.seed, "bounds": {"coarse": 1e-5, "fine_stage_on_identical_samples": 1e-5,
                                             "hypotheses": "indices bit-exact; 1e-5 on 99.99 %, 1e-3 on all"},
                  "worst": worst, "violations": violations}))
sys.exit(1 if violations else 0)
