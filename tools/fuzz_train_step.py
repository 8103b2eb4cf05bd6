"""Gradient campaign: the loss and all 48 parameter gradients of one optimisation step (run_plnerf.py:1283-1300: render_rays
of both networks, img2mse(rgb) + img2mse(rgb0), backward) against the CPU oracle's autograd on identical rays, targets and
pytest=True draws AND the path's own (detached) importance samples, over random configurations (sample counts, quadrature and
colour rules, background, density noise, disparity sampling, ragged ray counts 3-256), exact fp32 and f16x3.  Bounds per tensor: max error <= tol x max |g| of that
tensor with tol = 5e-4 / 2e-3 (fp32, coarse / fine network; the full-size tests' 2e-4 holds at 4096 rays -- at 3 rays one
density sample whose relu(sigma + noise) sits within rounding of zero is 3e-4 of a bias gradient) and 6e-3 / 3e-3 (f16x3: half
planes in the backward), the cosine of each network's whole gradient >= 0.999999 / 0.99999, the loss to 1e-5 -- those END-TO-END
numbers are reported (`worst`, `beyond_end_to_end_bounds`); what is BOUNDED since the third-seed pass of round 6 are the two stages of
tools/grad_stages.py on the path's own inputs: d loss / d raw against the fp64 oracle at the path's raw (1e-4 of its maximum), and the
parameter gradients against the fp64 oracle network's J^T g with the path's g_raw as cotangent (same tolerances; a tensor also passes
within 3x the fp32 oracle's own distance from fp64: a density bias at 3-256 rays is a sum that cancels to 1e-3 of its terms).
Test infrastructure (imports oracle/).  python tools/fuzz_train_step.py --cases 80 --seed 5 > out.json"""
import argparse, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import plnerf_amd as P
from oracle import plnerf_oracle as orc
from tools import grad_stages as GS

ap = argparse.ArgumentParser()
ap.add_argument("--cases", type=int, default=80)
ap.add_argument("--seed", type=int, default=5)
ap.add_argument("--precisions", default="fp32,f16x3")
a = ap.parse_args()
torch.set_num_threads(min(16, os.cpu_count() or 1))
dev = torch.device("cuda:0")
rng = np.random.default_rng(a.seed)
emb_fn, _ = P.get_embedder(10, 0)
embd_fn, _ = P.get_embedder(4, 0)
qfn = lambda inputs, viewdirs, fn: P.run_network(inputs, viewdirs, fn, emb_fn, embd_fn)
TOL = {"fp32": {"coarse": 5e-4, "fine": 2e-3, "cos": 0.999999}, "f16x3": {"coarse": 6e-3, "fine": 3e-3, "cos": 0.99999}}


sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from test_gpu_raygrad import _decisive_state_dict as decisive


def net(sd, precision):
    n = P.NeRF(D=8, W=256, input_ch=63, input_ch_views=27, output_ch=5, skips=[4], use_viewdirs=True, precision=precision)
    n.load_state_dict(sd)
    return n.to(dev)


worst = {p: {"loss": 0.0, "coarse": 0.0, "fine": 0.0, "cos_coarse": 1.0, "cos_fine": 1.0} for p in a.precisions.split(",")}
violations, cases, info = [], [], []
RMOD = sys.modules["plnerf_amd.render"]
for case in range(a.cases):
    # networks whose ReLU units are decisively on or off over the scene (tests/test_gpu_raygrad.py::_decisive_state_dict): with
    # ordinary weights 5 % of rows hold a unit within fp32 rounding of zero, and in a batch of a few thousand rows one flipped
    # unit is 1e-3 of a tensor's gradient (the second version of this campaign: fp32 coarse tensors 2e-4 ... 3e-2 off at
    # 3-256 rays, where the 4096-ray tests measure 1.5e-4)
    sharpen = True
    sd_c, sd_f = decisive(int(rng.integers(3))), decisive(3 + int(rng.integers(3)))
    mode = ["linear", "linear", "constant"][int(rng.integers(3))]
    kw = dict(N_samples=int(rng.choice([8, 17, 32, 64, 128])), N_importance=int(rng.choice([4, 9, 32, 64, 128])), mode=mode,
              color_mode=["midpoint", "left"][int(rng.integers(2))] if mode == "linear" else "midpoint",
              perturb=1.0, white_bkgd=bool(rng.integers(2)), raw_noise_std=float(rng.choice([0.0, 1.0])),
              lindisp=bool(rng.integers(2)), pytest=True)
    R = int(rng.choice([3, 33, 64, 130, 256]))
    batch, target = orc.synthetic_blender_rays(R, seed=5000 + case)
    rec = {"case": case, "R": R, "sharpened": sharpen, **{k: v for k, v in kw.items() if k != "pytest"}}
    for prec in worst:
        nc, nf = net(sd_c, prec), net(sd_f, prec)
        tap = {}
        RMOD.STAGE_TAP = tap
        try:
            ret = P.render_rays(batch.to(dev), nc, qfn, retraw=True, network_fine=nf, **kw)
        finally:
            RMOD.STAGE_TAP = None
        loss = P.img2mse(ret["rgb_map"], target.to(dev)) + P.img2mse(ret["rgb0"], target.to(dev))
        ret["raw"].retain_grad(); tap["raw0"].retain_grad()
        loss.backward()
        # The oracle's step ON THE PATH'S OWN IMPORTANCE SAMPLES (they are detached on both sides, run_plnerf.py:728, so no
        # gradient path changes): end to end a fine sample in another cdf bin moves a small batch's gradient by percents in
        # ANY two fp32 implementations (the first version of this campaign measured the same 4.6e-2 for the exact fp32
        # kernels and for f16x3) -- per stage is the meaningful statement.
        p_c = {k: v.clone().requires_grad_(True) for k, v in sd_c.items()}
        p_f = {k: v.clone().requires_grad_(True) for k, v in sd_f.items()}
        ref = orc.render_rays(batch, p_c, p_f, retraw=True, **kw)
        fs = orc.fine_stage(batch, p_f, tap["z_fine"].detach().cpu(), kw["mode"], kw["color_mode"], kw["white_bkgd"],
                            kw["raw_noise_std"], True, False)
        ref_loss = torch.mean((fs["rgb_map"] - target) ** 2) + torch.mean((ref["rgb0"] - target) ** 2)
        ref_loss.backward()
        g_c = {k: (v.grad if v.grad is not None else torch.zeros_like(v)) for k, v in p_c.items()}
        g_f = {k: (v.grad if v.grad is not None else torch.zeros_like(v)) for k, v in p_f.items()}
        e_loss = abs(float(loss.detach()) - float(ref_loss.detach()))
        out = {"loss": e_loss}
        bad = [] if e_loss <= 1e-5 else [f"loss {e_loss:.2e}"]
        for n_, grads, tag in ((nc, g_c, "coarse"), (nf, g_f, "fine")):
            e_w, which = 0.0, None
            g_max = max(float(grads[name].abs().max()) for name, _ in n_.named_parameters())
            for name, prm in n_.named_parameters():
                refg = grads[name]
                # (f16x3: a tensor whose entries are sums that cancel -- a bias at 3 rays -- is judged against a tenth of the
                # network's largest entry if that is more than its own: 2^-11 of the dz ENTRIES, tools/fuzz_mlp.py)
                scale = max(float(refg.abs().max()), 1e-9 if prec == "fp32" else 0.1 * g_max)
                e = float((prm.grad.cpu() - refg).abs().max()) / scale
                if e > e_w:
                    e_w, which = e, name
            fh = torch.cat([p.grad.cpu().double().reshape(-1) for _, p in n_.named_parameters()])
            fo = torch.cat([grads[name].double().reshape(-1) for name, _ in n_.named_parameters()])
            cos = float(torch.dot(fh, fo) / (fh.norm() * fo.norm())) if float(fo.norm()) > 0 else float(float(fh.norm()) == 0.0)
            out[tag], out["cos_" + tag], out["worst_tensor_" + tag] = e_w, cos, which
            if e_w > TOL[prec][tag]:
                bad.append(f"{tag} {which}: {e_w:.2e} of max |g|")
            if cos < TOL[prec]["cos"]:
                bad.append(f"{tag} cosine {cos:.7f}")
            w = worst[prec]
            w[tag], w["cos_" + tag] = max(w[tag], e_w), min(w["cos_" + tag], cos)
        worst[prec]["loss"] = max(worst[prec]["loss"], e_loss)
        rec[prec] = out
        if bad:
            info.append({"case": case, "precision": prec, "what": bad})
        # ---- the two stages, each on the path's own inputs (tools/grad_stages.py): these are the bounds
        b64 = batch.double()
        o, d, near, far, vd = b64[:, 0:3], b64[:, 3:6], b64[:, 6:7], b64[:, 7:8], batch[:, 8:11]
        staged = []
        for tag, n_, sd_, raw_t, z_t in (("coarse", nc, sd_c, tap["raw0"], tap["z_vals0"]), ("fine", nf, sd_f, ret["raw"], tap["z_fine"])):
            z = z_t.detach().cpu()
            rp = raw_t.detach().cpu().double().requires_grad_(True)
            rgb_p = orc.raw2outputs(rp, z.double(), near, far, d, kw["mode"], kw["color_mode"], kw["raw_noise_std"], True,
                                    kw["white_bkgd"], False)[0]
            torch.mean((rgb_p - target.double()) ** 2).backward()
            e_up = GS.upstream_error(raw_t.grad, rp.grad)
            pts = batch[:, None, 0:3] + batch[:, None, 3:6] * z[..., :, None]
            e_net, which_net, bad_net = GS.network_stage(orc.query_network, sd_, pts, vd, raw_t.grad,
                                                         {name: prm.grad for name, prm in n_.named_parameters()},
                                                         TOL[prec][tag], 0.0 if prec == "fp32" else 0.1)
            out["upstream_" + tag], out["network_" + tag], out["network_worst_" + tag] = e_up, e_net, which_net
            w = worst[prec]
            w["upstream_" + tag] = max(w.get("upstream_" + tag, 0.0), e_up)
            w["network_" + tag] = max(w.get("network_" + tag, 0.0), e_net)
            if e_up > GS.UP_TOL:
                staged.append(f"{tag} d loss / d raw: {e_up:.2e} of its maximum")
            staged += [f"{tag} network stage {b}" for b in bad_net]
        if staged:
            violations.append({"case": case, "precision": prec, "what": staged, "cfg": rec})
    cases.append(rec)
print(json.dumps({"what": "training-step gradient campaign vs the CPU oracle's autograd (pytest=True draws)", "cases": a.cases, "seed": a.seed,
                  "bounds": dict(TOL, upstream=GS.UP_TOL, stages="tools/grad_stages.py"), "worst": worst, "violations": violations,
                  "beyond_end_to_end_bounds": info,
                  "five_worst_fine": sorted(cases, key=lambda r: -max(r[p]["fine"] for p in worst))[:5]}))
sys.exit(1 if violations else 0)
