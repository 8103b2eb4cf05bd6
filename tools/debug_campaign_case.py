"""One case of tools/fuzz_train_step_depth.py again, with the oracle ALSO in fp64: per flagged tensor the path's value, the fp32
oracle's and the fp64 oracle's -- is a campaign violation the path's error or the fp32 yardstick's?
    python tools/debug_campaign_case.py --seed 132 --case 27 [--tensor alpha_linear.bias]"""
import argparse, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

ap = argparse.ArgumentParser()
ap.add_argument("--seed", type=int, default=132)
ap.add_argument("--case", type=int, default=27)
ap.add_argument("--tensor", default="alpha_linear.bias")
a = ap.parse_args()
# the campaign's own set-up code (networks, create_nerf): its source up to the case loop
src = open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "fuzz_train_step_depth.py")).read()
src = src[:src.index("for case in range(a.cases):")]
argv = sys.argv
sys.argv = ["fuzz_train_step_depth.py", "--cases", "0", "--seed", str(a.seed)]
G = {"__name__": "campaign_head", "__file__": os.path.join(os.path.dirname(os.path.abspath(__file__)), "fuzz_train_step_depth.py")}
try:
    exec(compile(src, "fuzz_train_step_depth.py", "exec"), G)
finally:
    sys.argv = argv
import plnerf_amd as P
from plnerf_amd import depth as Dp
from oracle import plnerf_oracle as orc
SDS, kws, W_SC, dev = G["SDS"], G["kws"], G["W_SC"], G["dev"]
rng = np.random.default_rng(a.seed)
for case in range(a.case + 1):
    s_c, s_f = int(rng.integers(2)), 2 + int(rng.integers(2))
    cfg = dict(N_samples=int(rng.choice([8, 17, 32, 64])), N_importance=int(rng.choice([4, 9, 32, 64])), mode="linear",
               color_mode=["midpoint", "left"][int(rng.integers(2))], white_bkgd=bool(rng.integers(2)),
               raw_noise_std=float(rng.choice([0.0, 1.0])), perturb=1.0)
    R = int(rng.choice([3, 33, 64, 130]))
batch, target = orc.synthetic_blender_rays(R, seed=13000 + a.case)
gen = torch.Generator().manual_seed(13000 + a.case)
target_h = 2.0 + 4.0 * torch.rand(3, R, 1, generator=gen)
near, far = batch[:, 6:7], batch[:, 7:8]
print(json.dumps({"case": a.case, "R": R, **cfg}))


def oracle(z_fine, u, dt):
    p_c = {k: v.to(dt).clone().requires_grad_(True) for k, v in SDS[s_c].items()}
    p_f = {k: v.to(dt).clone().requires_grad_(True) for k, v in SDS[s_f].items()}
    b, t, th = batch.to(dt), target.to(dt), target_h.to(dt)
    ref = orc.render_rays_depth(b, p_c, p_f, cfg["N_samples"], "linear", cfg["color_mode"], perturb=1.0,
                                N_importance=cfg["N_importance"], white_bkgd=cfg["white_bkgd"], raw_noise_std=cfg["raw_noise_std"],
                                pytest=True)
    fs = orc.fine_stage(b, p_f, z_fine.to(dt), "linear", cfg["color_mode"], cfg["white_bkgd"], cfg["raw_noise_std"], True,
                        depth_variant=True)
    fs["raw"].retain_grad()
    hyp = orc.sample_pdf_reformulation(z_fine.to(dt), fs["weights"], fs["tau"], fs["T"], near.to(dt), far.to(dt), cfg["N_importance"],
                                       u=u.to(dt))[0]
    loss = torch.mean((fs["rgb_map"] - t) ** 2) + W_SC * orc.compute_space_carving_loss(hyp, th) + torch.mean((ref["rgb0"] - t) ** 2)
    loss.backward()
    return {k: v.grad for k, v in p_f.items()}, {k: v.grad for k, v in p_c.items()}, float(loss.detach()), hyp.detach(), fs["raw"].grad.detach(), fs["raw"].detach()


for prec, kw in kws.items():
    kw["network_fn"].load_state_dict(SDS[s_c]); kw["network_fine"].load_state_dict(SDS[s_f])
    kw["network_fn"].zero_grad(); kw["network_fine"].zero_grad()
    ret = Dp.render_rays(batch.to(dev), retraw=True, pytest=True, **dict(kw, **cfg))
    sc = Dp.compute_space_carving_loss(ret["pred_hyp"], target_h.to(dev))
    loss = P.img2mse(ret["rgb_map"], target.to(dev)) + W_SC * sc + P.img2mse(ret["rgb0"], target.to(dev))
    ret["raw"].retain_grad()
    loss.backward()
    z_fine, u = ret["z_vals"].detach().cpu(), ret["u"].detach().cpu()
    g32, _, l32, h32, gr32, raw32 = oracle(z_fine, u, torch.float32)
    g64, _, l64, h64, gr64, raw64 = oracle(z_fine, u, torch.float64)
    gr = ret["raw"].grad.detach().cpu().double()
    print(f"   d loss / d raw (fine): max |g| {float(gr64.abs().max()):.3e}; path vs fp64 {float((gr - gr64).abs().max()):.3e}, "
          f"fp32 oracle vs fp64 {float((gr32.double() - gr64).abs().max()):.3e}; sums of the sigma column: path {float(gr[..., 3].sum()):.6e} "
          f"fp64 {float(gr64[..., 3].sum()):.6e}; sum |.| {float(gr64[..., 3].abs().sum()):.3e}; raw: path vs fp64 "
          f"{float((ret['raw'].detach().cpu().double() - raw64).abs().max()):.3e}")
    # the SAME question with the conditioning taken out: the fp64 oracle's d loss / d raw AT THE PATH'S OWN raw (what an exact
    # backward of the path's forward values would return)
    rp = ret["raw"].detach().cpu().double().requires_grad_(True)
    b64 = batch.double()
    rgb_p, _, _, w_p, _, tau_p, T_p = orc.raw2outputs(rp, z_fine.double(), b64[:, 6:7], b64[:, 7:8], b64[:, 3:6], "linear", cfg["color_mode"],
                                                      cfg["raw_noise_std"], True, cfg["white_bkgd"], False)
    hyp_p = orc.sample_pdf_reformulation(z_fine.double(), w_p, tau_p, T_p, b64[:, 6:7], b64[:, 7:8], cfg["N_importance"], u=u.double())[0]
    (torch.mean((rgb_p - target.double()) ** 2) + W_SC * orc.compute_space_carving_loss(hyp_p, target_h.double())).backward()
    print(f"   fp64 oracle's d loss / d raw AT THE PATH'S raw: path vs that {float((gr - rp.grad).abs().max()):.3e}; that vs the fp64 oracle's "
          f"own {float((rp.grad - gr64).abs().max()):.3e}; sigma-column sums: {float(rp.grad[..., 3].sum()):.6e} (path {float(gr[..., 3].sum()):.6e})")
    # a hypothesis whose u sits within rounding of a cdf knot lands in another bin in one of the two evaluations: the gradient
    # of that ray then belongs to another interval -- not an arithmetic error of either side
    ph = ret["pred_hyp"].detach().cpu().double()
    d = (ph - h64.reshape(ph.shape)).abs()
    print(f"   hypotheses: {ph.numel()}, max |path - fp64 oracle| {float(d.max()):.3e}, beyond 1e-4: {int((d > 1e-4).sum())} "
          f"(rays {sorted(set((d > 1e-4).nonzero()[:, 0].tolist()))}); fp32 oracle vs fp64: beyond 1e-4: "
          f"{int(((h32.double().reshape(ph.shape) - h64.reshape(ph.shape)).abs() > 1e-4).sum())}")
    g_max = max(float(x.abs().max()) for x in g64.values() if x is not None)
    print(f"== {prec}: loss path {float(loss):.8f} oracle fp32 {l32:.8f} fp64 {l64:.8f}; network's largest |g| {g_max:.3e}")
    rows = []
    for name, prm in kw["network_fine"].named_parameters():
        r64 = g64[name]
        scale = max(float(r64.abs().max()), 0.1 * g_max)
        e_path = float((prm.grad.cpu().double() - r64).abs().max()) / scale
        e_o32 = float((g32[name].double() - r64).abs().max()) / scale
        rows.append((e_path, e_o32, name, float(r64.abs().max())))
    for e_path, e_o32, name, own in sorted(rows, reverse=True)[:6]:
        print(f"   fine {name:28s} path vs fp64 {e_path:.2e}   fp32 oracle vs fp64 {e_o32:.2e}   own max |g| {own:.3e}")
    if a.tensor:
        prm = dict(kw["network_fine"].named_parameters())[a.tensor]
        print(f"   {a.tensor}: path {prm.grad.cpu().flatten()[:4].tolist()}  fp32 oracle {g32[a.tensor].flatten()[:4].tolist()}  "
              f"fp64 oracle {g64[a.tensor].flatten()[:4].tolist()}")
