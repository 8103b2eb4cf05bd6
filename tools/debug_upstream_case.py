"""One case of tools/fuzz_train_step_depth.py, its upstream stage only: where does the path's d loss / d raw (fine pass) differ
from the fp64 oracle's at the path's own raw, and what does the sampler see there?
    python tools/debug_upstream_case.py --seed 132 --case 65 --precision fp32"""
import argparse, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

ap = argparse.ArgumentParser()
ap.add_argument("--seed", type=int, default=132)
ap.add_argument("--case", type=int, default=65)
ap.add_argument("--precision", default="fp32")
a = ap.parse_args()
here = os.path.dirname(os.path.abspath(__file__))
src = open(os.path.join(here, "fuzz_train_step_depth.py")).read()
src = src[:src.index("violations, info = [], []")]
argv = sys.argv
sys.argv = ["fuzz_train_step_depth.py", "--cases", "0", "--seed", str(a.seed)]
G = {"__name__": "campaign_head", "__file__": os.path.join(here, "fuzz_train_step_depth.py")}
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
print(json.dumps({"case": a.case, "R": R, **cfg}))
kw = kws[a.precision]
kw["network_fn"].load_state_dict(SDS[s_c]); kw["network_fine"].load_state_dict(SDS[s_f])
kw["network_fn"].zero_grad(); kw["network_fine"].zero_grad()
ret = Dp.render_rays(batch.to(dev), retraw=True, pytest=True, **dict(kw, **cfg))
sc = Dp.compute_space_carving_loss(ret["pred_hyp"], target_h.to(dev))
loss = P.img2mse(ret["rgb_map"], target.to(dev)) + W_SC * sc + P.img2mse(ret["rgb0"], target.to(dev))
ret["raw"].retain_grad(); ret["pred_hyp"].retain_grad()
loss.backward()
z = ret["z_vals"].detach().cpu()
u = ret["u"].detach().cpu()
up, parts = {}, {}
for dt in (torch.float64, torch.float32):
    for part in ("rgb", "hyp"):
        rp = ret["raw"].detach().cpu().to(dt).requires_grad_(True)
        bd, td = batch.to(dt), target.to(dt)
        rgb_p, _, _, w_p, _, tau_p, T_p = orc.raw2outputs(rp, z.to(dt), bd[:, 6:7], bd[:, 7:8], bd[:, 3:6], "linear", cfg["color_mode"],
                                                          cfg["raw_noise_std"], True, cfg["white_bkgd"], False)
        if part == "rgb":
            torch.mean((rgb_p - td) ** 2).backward()
        else:
            hyp_p, T0, tau0, s0, inds = orc.sample_pdf_reformulation(z.to(dt), w_p, tau_p, T_p, bd[:, 6:7], bd[:, 7:8], cfg["N_importance"],
                                                                     u=u.to(dt), return_inds=True)
            (W_SC * orc.compute_space_carving_loss(hyp_p, target_h.to(dt))).backward()
            if dt == torch.float64:
                tau64, T64, inds64, hyp64 = tau_p.detach(), T_p.detach(), inds, hyp_p.detach()
        parts[(dt, part)] = rp.grad.double()
    up[dt] = parts[(dt, "rgb")] + parts[(dt, "hyp")]
g = ret["raw"].grad.detach().cpu().double()
gm = float(up[torch.float64].abs().max())
e = (g - up[torch.float64]).abs()
e32 = (up[torch.float32] - up[torch.float64]).abs()
print(f"max |g_raw| {gm:.3e}; path vs fp64 {float(e.max()) / gm:.3e}; fp32 oracle vs fp64 {float(e32.max()) / gm:.3e}; "
      f"hypothesis part alone: max {float(parts[(torch.float64, 'hyp')].abs().max()):.3e}, fp32 oracle's distance {float((parts[(torch.float32, 'hyp')] - parts[(torch.float64, 'hyp')]).abs().max()) / gm:.3e}; "
      f"rgb part: fp32 oracle's distance {float((parts[(torch.float32, 'rgb')] - parts[(torch.float64, 'rgb')]).abs().max()) / gm:.3e}")
flat = torch.argsort(e.reshape(-1), descending=True)[:6]
for f in flat.tolist():
    r, s_, c = f // (e.shape[1] * 4), (f // 4) % e.shape[1], f % 4
    print(f"  ray {r} sample {s_} channel {c}: path {float(g[r, s_, c]):+.6e} fp64 {float(up[torch.float64][r, s_, c]):+.6e} fp32 oracle {float(up[torch.float32][r, s_, c]):+.6e}"
          f"   (hyp part fp64 {float(parts[(torch.float64, 'hyp')][r, s_, c]):+.3e})")
r = int(flat[0]) // (e.shape[1] * 4)
# the hypotheses of that ray: which of the closed form's guards (run_nerf_helpers.py:341-359: max(eps, .) and the final clamp) and
# branch thresholds sit within rounding of their switch?  On either side of a guard the VALUE is continuous but the gradient is not.
eps, zt = 1e-3, 1e-4
knots = torch.cat([batch[r, 6:7].double(), z[r].double(), batch[r, 7:8].double()])
print(f"  ray {r}: guards within 1e-4 (relative) of their switch, per hypothesis")
for j, b in enumerate(inds64[r].tolist()):
    lo = max(b - 1, 0); hi = min(b, tau64.shape[1] - 1)
    tl, tr, Tl, uu = float(tau64[r, lo]), float(tau64[r, hi]), float(T64[r, lo]), float(u[r, j])
    span = float(knots[hi] - knots[lo]); d = tr - tl
    q = (1 - uu) / max(eps, Tl)
    ln = -np.log(max(eps, q))
    disc = tl * tl + 2 * d * ln / max(eps, span) if d > 0 else tl * tl + 2 * (tr - tl) * ln / max(eps, span)
    t = span * (-tl + np.sqrt(max(eps, disc))) / max(eps, d) if d >= zt else (span * (tl - np.sqrt(max(eps, disc))) / max(eps, -d) if d <= -zt else 0.0)
    near_ = []
    for name, val, thr in (("T_l|eps", Tl, eps), ("(1-u)/T|eps", q, eps), ("disc|eps", disc, eps), ("|dtau||eps", abs(d), eps), ("|dtau||zt", abs(d), zt),
                           ("span|eps", span, eps), ("t|eps", t, eps), ("t|span", t, span)):
        if abs(val - thr) <= 1e-4 * max(abs(thr), 1e-30) or (name in ("t|eps", "t|span") and ((name == "t|eps" and t < eps) or (name == "t|span" and t > span)) and abs(d) >= zt):
            near_.append(f"{name}: {val:.7e}")
    dh = float(ret["pred_hyp"][r, j]) - float(hyp64[r, j])
    if near_ or abs(dh) > 1e-4:
        print(f"    hyp {j}: bin {lo} dtau {d:+.4e} T_l {Tl:.4e} u {uu:.6f} t {t:.5e} span {span:.5e}  path - oracle {dh:+.2e}  {near_}")

# ---- the sampler's backward alone on that ray: plnerf_sample_pl_bwd on the fp32 cast of the fp64 oracle's (tau, T, weights) against fp64 autograd
from plnerf_amd import functional as Fn
rp = ret["raw"].detach().cpu().double()
b64 = batch.double()
_, _, _, w64, _, tau_o, T_o = orc.raw2outputs(rp, z.double(), b64[:, 6:7], b64[:, 7:8], b64[:, 3:6], "linear", cfg["color_mode"], cfg["raw_noise_std"], True,
                                              cfg["white_bkgd"], False)
cot = torch.zeros(R, cfg["N_importance"], dtype=torch.float64); cot[r] = 1.0


def sgrads(dt):
    tr, Tr_ = tau_o.to(dt).clone().requires_grad_(True), T_o.to(dt).clone().requires_grad_(True)
    s_ref, _, _, _, ii = orc.sample_pdf_reformulation(z.to(dt), w64.to(dt), tr, Tr_, b64[:, 6:7].to(dt), b64[:, 7:8].to(dt), cfg["N_importance"], u=u.to(dt),
                                                      return_inds=True)
    (s_ref * cot.to(dt)).sum().backward()
    return tr.grad[r].double(), Tr_.grad[r].double(), ii[r], s_ref[r].detach().double()
g64t, g64T, i64, s64 = sgrads(torch.float64)
g32t, g32T, i32, s32 = sgrads(torch.float32)
gd = lambda t: t.float().to(dev).contiguous()
tau_h, T_h = gd(tau_o).requires_grad_(True), gd(T_o).requires_grad_(True)
s_hip = Fn.sample_pl(gd(z), gd(w64), tau_h, T_h, gd(b64[:, 6:7]), gd(b64[:, 7:8]), gd(u), 1e-4, 1e-3)
(s_hip * gd(cot)).sum().backward()
ht, hT = tau_h.grad[r].cpu().double(), T_h.grad[r].cpu().double()
print(f"  sampler alone, ray {r}: bins fp64 == fp32 oracle: {bool((i64 == i32).all())}; samples HIP - fp64 max {float((s_hip[r].detach().cpu().double() - s64).abs().max()):.2e}")
print(f"    g_tau: max {float(g64t.abs().max()):.3e}; HIP - fp64 {float((ht - g64t).abs().max()):.3e}; fp32 oracle - fp64 {float((g32t - g64t).abs().max()):.3e}")
print(f"    g_T:   max {float(g64T.abs().max()):.3e}; HIP - fp64 {float((hT - g64T).abs().max()):.3e}; fp32 oracle - fp64 {float((g32T - g64T).abs().max()):.3e}")
k = int((hT - g64T).abs().argmax())
print(f"    worst g_T knot {k}: HIP {float(hT[k]):+.6e} fp64 {float(g64T[k]):+.6e} fp32 oracle {float(g32T[k]):+.6e}; hypotheses in that bin: "
      f"{[(j, round(float(u[r, j]), 6), round(float(s64[j]), 6), round(float(s_hip[r, j]), 6)) for j in range(cfg['N_importance']) if max(int(i64[j]) - 1, 0) == k]}")
print(f"    knot {k}: T {float(T_o[r, k]):.7e} tau_l {float(tau_o[r, k]):.7e} tau_r {float(tau_o[r, min(k + 1, tau_o.shape[1] - 1)]):.7e}")

# ---- the path's own (tau, T, bins) of that ray against the oracle's at the same raw (depth.STAGE_TAP: the separate launches, bit-identical to the fused ones)
TAPMOD = sys.modules["plnerf_amd.depth"]
tap = {}
TAPMOD.STAGE_TAP = tap
try:
    with torch.no_grad():
        ret2 = Dp.render_rays(batch.to(dev), retraw=True, pytest=True, **dict(kw, **cfg))
finally:
    TAPMOD.STAGE_TAP = None
tp, Tp, ip = tap["tau"][r].cpu().double(), tap["T"][r].cpu().double(), tap["hyp_inds"][r].cpu()
print(f"  path's own stage values, ray {r}: raw identical to the first run: {bool((ret2['raw'] == ret['raw']).all())}; bins equal the oracle's: {bool((ip == i64).all())} "
      f"(differing hypotheses {[(j, int(ip[j]), int(i64[j])) for j in range(len(ip)) if int(ip[j]) != int(i64[j])]})")
print(f"    tau: max |path - oracle| {float((tp[1:-1] - tau_o[r, 1:-1]).abs().max()):.3e}; T: max rel {float(((Tp - T_o[r]).abs() / T_o[r].clamp(min=1e-30)).max()):.3e}")

# ---- d loss / d hypothesis: the space-carving term's own gradient (its min over the target hypotheses is another switch)
hp = ret["pred_hyp"].detach().cpu().double().requires_grad_(True)
(W_SC * orc.compute_space_carving_loss(hp, target_h.double())).backward()
gh = ret["pred_hyp"].grad.detach().cpu().double()
dh_ = (gh - hp.grad).abs()
print(f"  d loss / d hypothesis: max |g| {float(hp.grad.abs().max()):.3e}; path - fp64 oracle (at the path's hypotheses) max {float(dh_.max()):.3e} "
      f"at ray {int(dh_.max(1).values.argmax())}; on ray {r}: {float(dh_[r].max()):.3e}, hypotheses beyond 1e-9: "
      f"{[(j, float(gh[r, j]), float(hp.grad[r, j])) for j in range(gh.shape[1]) if float(dh_[r, j]) > 1e-9][:6]}")

# ---- the quadrature's backward alone: cotangents (g_tau, g_T) = the fp64 sampler gradients of the space-carving term, into plnerf_quad_bwd and into fp64 autograd
tr, Tr_ = tau_o.clone().requires_grad_(True), T_o.clone().requires_grad_(True)
s_all = orc.sample_pdf_reformulation(z.double(), w64, tr, Tr_, b64[:, 6:7], b64[:, 7:8], cfg["N_importance"], u=u.double())[0]
(W_SC * orc.compute_space_carving_loss(s_all, target_h.double())).backward()
Gt, GT = tr.grad, Tr_.grad
rq = ret["raw"].detach().cpu().double().requires_grad_(True)
o64 = orc.raw2outputs(rq, z.double(), b64[:, 6:7], b64[:, 7:8], b64[:, 3:6], "linear", cfg["color_mode"], cfg["raw_noise_std"], True, cfg["white_bkgd"], False)
((o64[5] * Gt).sum() + (o64[6] * GT).sum()).backward()
rh = ret["raw"].detach().clone().requires_grad_(True)
oh = P.raw2outputs(rh, gd(z), gd(b64[:, 6:7]), gd(b64[:, 7:8]), gd(b64[:, 3:6]), "linear", cfg["color_mode"], raw_noise_std=cfg["raw_noise_std"], pytest=True,
                   white_bkgd=cfg["white_bkgd"])
((oh[5] * gd(Gt)).sum() + (oh[6] * gd(GT)).sum()).backward()
eq = (rh.grad.cpu().double() - rq.grad).abs()
print(f"  quadrature backward alone (cotangents g_tau max {float(Gt.abs().max()):.3e}, g_T max {float(GT.abs().max()):.3e}): d / d raw max {float(rq.grad.abs().max()):.3e}; "
      f"HIP - fp64 max {float(eq.max()):.3e} at ray {int(eq.reshape(R, -1).max(1).values.argmax())}; on ray {r}: {float(eq[r].max()):.3e}")

# ---- the backward's predicates (which side of each guard), in fp32 as the kernel evaluates them, on the path's (tau, T) and on the oracle's
f32 = np.float32


def predicates(tau_v, T_v, j):
    b = int(i64[j]); lo_ = max(b - 1, 0); hi_ = min(b, len(tau_v) - 1)
    s0_, s1_ = f32(knots[lo_]), f32(knots[hi_])
    di = min(lo_, len(tau_v) - 2)
    d_ = f32(tau_v[di + 1]) - f32(tau_v[di])
    if not (d_ >= f32(zt) or d_ <= -f32(zt)):
        return ("flat",)
    rising = d_ >= f32(zt)
    a0_, a1_, T0_, e_ = f32(tau_v[lo_]), f32(tau_v[hi_]), f32(T_v[lo_]), f32(eps)
    L_ = f32(s1_ - s0_)
    ratio = f32(f32(1.0) - f32(u[r, j])) / max(e_, T0_)
    ln_ = -np.log(max(e_, f32(ratio)), dtype=f32)
    span_ = max(e_, L_)
    q_ = f32(f32(f32(2.0) * (a1_ - a0_ if rising else a0_ - a1_)) * ln_) / span_
    disc_ = f32(a0_ * a0_ + q_) if rising else f32(a0_ * a0_ - q_)
    sq_ = np.sqrt(max(e_, disc_), dtype=f32)
    diff_ = a1_ - a0_ if rising else a0_ - a1_
    den_ = max(e_, diff_)
    t_ = f32(L_ * (f32(-a0_ + sq_) if rising else f32(a0_ - sq_))) / den_
    return ("rising" if rising else "falling", bool(t_ >= e_ and t_ <= L_), bool(disc_ > e_), bool(diff_ > e_), bool(ratio > e_ and T0_ > e_), float(t_), float(L_))
for j in range(cfg["N_importance"]):
    pa, pb = predicates(tp.numpy(), Tp.numpy(), j), predicates(tau_o[r].numpy(), T_o[r].numpy(), j)
    if pa[:5] != pb[:5]:
        print(f"    hypothesis {j} (bin {max(int(i64[j]) - 1, 0)}): path's inputs {pa}   oracle's inputs {pb}")

# ---- the sampler's backward on the PATH's own (tau, T), one hypothesis of that ray at a time: where does the kernel's gradient leave fp64 autograd's?
tau_pp, T_pp = tap["tau"].detach().cpu().double(), tap["T"].detach().cpu().double()
w_pp = tap["weights_full"].detach().cpu().double()
for j in range(cfg["N_importance"]):
    cotj = torch.zeros(R, cfg["N_importance"], dtype=torch.float64); cotj[r, j] = 1.0
    trj, Trj = tau_pp.clone().requires_grad_(True), T_pp.clone().requires_grad_(True)
    sj = orc.sample_pdf_reformulation(z.double(), w_pp, trj, Trj, b64[:, 6:7], b64[:, 7:8], cfg["N_importance"], u=u.double())[0]
    (sj * cotj).sum().backward()
    th, Th = gd(tau_pp).requires_grad_(True), gd(T_pp).requires_grad_(True)
    sh = Fn.sample_pl(gd(z), gd(w_pp), th, Th, gd(b64[:, 6:7]), gd(b64[:, 7:8]), gd(u), 1e-4, 1e-3)
    (sh * gd(cotj)).sum().backward()
    et = float((th.grad[r].cpu().double() - trj.grad[r]).abs().max()); eT = float((Th.grad[r].cpu().double() - Trj.grad[r]).abs().max())
    mt, mT = float(trj.grad[r].abs().max()), float(Trj.grad[r].abs().max())
    if et > 1e-4 * max(mt, 1e-30) + 1e-9 or eT > 1e-4 * max(mT, 1e-30) + 1e-9:
        print(f"    hypothesis {j}: bin {max(int(i64[j]) - 1, 0)} u {float(u[r, j]):.6f}: g_tau HIP - fp64 {et:.3e} of {mt:.3e}; g_T {eT:.3e} of {mT:.3e}; sample HIP {float(sh[r, j]):.7f} fp64 {float(sj[r, j]):.7f}; "
              f"branch codes {predicates(tp.numpy(), Tp.numpy(), j)}")

# ---- the quadrature's backward with ALL the step's cotangents at once (image term's g_rgb + the sampler's g_tau, g_T)
G_rgb = (2.0 / (R * 3)) * (o64[0].detach() - target.double())
rq2 = ret["raw"].detach().cpu().double().requires_grad_(True)
o2 = orc.raw2outputs(rq2, z.double(), b64[:, 6:7], b64[:, 7:8], b64[:, 3:6], "linear", cfg["color_mode"], cfg["raw_noise_std"], True, cfg["white_bkgd"], False)
((o2[0] * G_rgb).sum() + (o2[5] * Gt).sum() + (o2[6] * GT).sum()).backward()
rh2 = ret["raw"].detach().clone().requires_grad_(True)
oh2 = P.raw2outputs(rh2, gd(z), gd(b64[:, 6:7]), gd(b64[:, 7:8]), gd(b64[:, 3:6]), "linear", cfg["color_mode"], raw_noise_std=cfg["raw_noise_std"], pytest=True,
                    white_bkgd=cfg["white_bkgd"])
((oh2[0] * gd(G_rgb)).sum() + (oh2[5] * gd(Gt)).sum() + (oh2[6] * gd(GT)).sum()).backward()
eq2 = (rh2.grad.cpu().double() - rq2.grad).abs()
print(f"  quadrature backward, all cotangents: HIP - fp64 max {float(eq2.max()):.3e}; on ray {r}: {float(eq2[r].max()):.3e}; "
      f"this fp64 composition vs the oracle's end-to-end fp64 on ray {r}: {float((rq2.grad[r] - up[torch.float64][r]).abs().max()):.3e}; "
      f"in-situ path vs this HIP composition on ray {r}: {float((g[r] - rh2.grad[r].cpu().double()).abs().max()):.3e}")

# ---- the same step through the SEPARATE launches (STAGE_TAP set) against the fused fine epilogue of the first run
tap3 = {}
TAPMOD.STAGE_TAP = tap3
try:
    kw["network_fn"].zero_grad(); kw["network_fine"].zero_grad()
    ret3 = Dp.render_rays(batch.to(dev), retraw=True, pytest=True, **dict(kw, **cfg))
finally:
    TAPMOD.STAGE_TAP = None
loss3 = P.img2mse(ret3["rgb_map"], target.to(dev)) + W_SC * Dp.compute_space_carving_loss(ret3["pred_hyp"], target_h.to(dev)) + P.img2mse(ret3["rgb0"], target.to(dev))
ret3["raw"].retain_grad()
loss3.backward()
g3 = ret3["raw"].grad.detach().cpu().double()
print(f"  separate launches vs fused: pred_hyp identical {bool((ret3['pred_hyp'] == ret['pred_hyp']).all())}; d loss / d raw max diff {float((g3 - g).abs().max()):.3e} "
      f"(ray {int((g3 - g).abs().reshape(R, -1).max(1).values.argmax())}); separate vs fp64 on ray {r}: {float((g3[r] - up[torch.float64][r]).abs().max()):.3e}; fused vs fp64: {float((g[r] - up[torch.float64][r]).abs().max()):.3e}")

# ---- the sampler's backward on the path's own (tau, T) with the step's actual cotangent (all hypotheses at once)
cot_all = gh.clone()
trA, TrA = tau_pp.clone().requires_grad_(True), T_pp.clone().requires_grad_(True)
sA = orc.sample_pdf_reformulation(z.double(), w_pp, trA, TrA, b64[:, 6:7], b64[:, 7:8], cfg["N_importance"], u=u.double())[0]
(sA * cot_all).sum().backward()
thA, ThA = gd(tau_pp).requires_grad_(True), gd(T_pp).requires_grad_(True)
shA = Fn.sample_pl(gd(z), gd(w_pp), thA, ThA, gd(b64[:, 6:7]), gd(b64[:, 7:8]), gd(u), 1e-4, 1e-3)
(shA * gd(cot_all)).sum().backward()
dt_ = (thA.grad.cpu().double() - trA.grad).abs(); dT_ = (ThA.grad.cpu().double() - TrA.grad).abs()
print(f"  sampler backward, the step's own cotangent: g_tau max {float(trA.grad.abs().max()):.3e}, HIP - fp64 max {float(dt_.max()):.3e} (ray {int(dt_.max(1).values.argmax())}), on ray {r}: {float(dt_[r].max()):.3e} at knot {int(dt_[r].argmax())}; "
      f"g_T max {float(TrA.grad.abs().max()):.3e}, HIP - fp64 {float(dT_.max()):.3e}, on ray {r}: {float(dT_[r].max()):.3e} at knot {int(dT_[r].argmax())}")
kk = int(dt_[r].argmax())
print(f"    knot {kk}: g_tau HIP {float(thA.grad[r, kk]):+.6e} fp64 {float(trA.grad[r, kk]):+.6e}; hypotheses touching it: "
      f"{[(j, max(int(i64[j]) - 1, 0), min(int(i64[j]), tau_pp.shape[1] - 1), float(cot_all[r, j])) for j in range(cfg['N_importance']) if kk in (max(int(i64[j]) - 1, 0), min(int(i64[j]), tau_pp.shape[1] - 1))]}")

# ---- fp64 sampler gradients at the path's (tau, T) against fp64 at the oracle's (tau, T): the same function, inputs 4e-7 apart
d1, d2 = (trA.grad[r] - Gt[r]).abs(), (TrA.grad[r] - GT[r]).abs()
print(f"  fp64 at the path's inputs vs fp64 at the oracle's, ray {r}: g_tau max diff {float(d1.max()):.3e} at knot {int(d1.argmax())} (of {float(Gt[r].abs().max()):.3e}); g_T {float(d2.max()):.3e} at knot {int(d2.argmax())} (of {float(GT[r].abs().max()):.3e})")
kn = torch.cat([b64[:, 6:7], z.double(), b64[:, 7:8]], -1)
from tools import grad_stages as GS


def bins_of(w_):
    cdf_ = torch.cat([torch.zeros(R, 1, dtype=torch.float64), torch.cumsum(w_, -1)], -1)
    cdf_[:, -1] = 1.0
    return torch.searchsorted(cdf_, u.double().contiguous(), right=True)
cP = GS._sampler_branch_code(kn, tau_pp, T_pp, u.double(), bins_of(w_pp), 1e-3, 1e-4)
cO = GS._sampler_branch_code(kn, tau_o, T_o, u.double(), bins_of(w64), 1e-3, 1e-4)
print(f"    branch codes differ on hypotheses {[(j, int(cP[r, j]), int(cO[r, j])) for j in range(cfg['N_importance']) if int(cP[r, j]) != int(cO[r, j])]}")
for kx in sorted(set([int(d1.argmax()), int(d2.argmax())])):
    print(f"    knot {kx}: tau path {float(tau_pp[r, kx]):.9e} oracle {float(tau_o[r, kx]):.9e}; T path {float(T_pp[r, kx]):.9e} oracle {float(T_o[r, kx]):.9e}; "
          f"hypotheses in bin: {[(j, float(u[r, j]), float(sA[r, j]), float(s_all[r, j])) for j in range(cfg['N_importance']) if max(int(i64[j]) - 1, 0) == kx]}")
