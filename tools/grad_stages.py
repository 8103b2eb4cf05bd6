"""The training step's gradient, judged stage by stage (shared by fuzz_train_step.py and fuzz_train_step_depth.py).

A parameter gradient is J_net^T g_raw with g_raw = d loss / d raw.  Two statements, each on the path's OWN inputs:

  upstream   g_raw of the path (plnerf_quad_bwd, plnerf_sample_pl_bwd, the loss kernels) against the fp64 oracle's d loss / d raw
             evaluated AT THE PATH'S raw -- what an exact backward of the path's forward values returns (depth-supervised step:
             or within 3x the fp32 oracle's own distance from fp64 at that raw -- the sampler's closed-form gradient cancels);
  network    the path's parameter gradients against the fp64 oracle network's J^T g with the path's g_raw as cotangent, and
             the fp32 oracle network's distance from the same fp64 yardstick: a tensor passes inside its tolerance OR
             within 3x the fp32 reference's own rounding (a density bias is a sum that cancels: two fp32 summations of it
             differ by 1e-3 of its value while every term agrees to 1e-7).

What the end-to-end comparison with the oracle's autograd adds on top is the forward's own difference (inside the 1e-5
contract) seen through the conditioning of the LOSS: the sampler's closed form carries 1 / (tau_r - tau_l)^2 terms, and a
1e-6 difference in a density moves d loss / d raw by 5e-5 of its maximum on rays whose interval sits near the zero
threshold (third-seed campaign, case 27: the path's f16x3 g_raw is 1.5e-7 from the fp64 oracle's AT ITS OWN raw and
4.9e-5 from the oracle's at the oracle's raw; a density bias -- the sum of that column -- then reads 8 % off).  The
campaigns report those end-to-end numbers; they bound the two stages.  Test infrastructure (imports oracle/)."""
import torch

UP_TOL = 1e-4      # upstream gradient: max error / max |g_raw| (measured 1e-5 ... 3e-5)


def upstream_error(g_path, g_ref, keep=None):
    """max |g_path - g_ref| / max |g_ref| of two d loss / d raw tensors [R, S, 4] (fp64 on the CPU); keep: bool [R], the rays that count."""
    e = (g_path.double().cpu() - g_ref.double()).abs()
    if keep is not None:
        e = e[keep]
    return (float(e.max()) if e.numel() else 0.0) / max(float(g_ref.abs().max()), 1e-30)


def _sampler_branch_code(knots, tau, T, u, inds, eps, zt):
    """Per hypothesis, which side of every switch of invert_segment / its gradient the evaluation stands on (csrc/sampler.hip:
    invert_segment_grad's predicate block, op for op, in the dtype of the arguments): 0 = flat interval; 1 / 2 = rising / falling
    and clamped (no gradient); else rising / falling + which of the discriminant, the slope and the log term are live."""
    K = knots.shape[-1]
    below = torch.clamp(inds - 1, min=0)
    above = torch.clamp(inds, max=K - 1)
    s0, s1 = torch.gather(knots, -1, below), torch.gather(knots, -1, above)
    T0 = torch.gather(T, -1, below)
    tau0, tau1 = torch.gather(tau, -1, below), torch.gather(tau, -1, above)
    d = torch.gather(tau[..., 1:] - tau[..., :-1], -1, torch.clamp(below, max=K - 2))
    rising, falling = d >= zt, d <= -zt
    e = torch.full_like(T0, eps)
    L = s1 - s0
    ratio = (1 - u) / torch.max(e, T0)
    ln = -torch.log(torch.max(e, ratio))
    span = torch.max(e, L)
    diff = torch.where(rising, tau1 - tau0, tau0 - tau1)
    q = (2 * diff * ln) / span
    disc = torch.where(rising, tau0 * tau0 + q, tau0 * tau0 - q)
    sq = torch.sqrt(torch.max(e, disc))
    t = (L * torch.where(rising, -tau0 + sq, tau0 - sq)) / torch.max(e, diff)
    flows = (t >= e) & (t <= L)
    base = rising.long() + 2 * falling.long()
    live = 4 + 8 * (disc > e).long() + 16 * (diff > e).long() + 32 * ((ratio > e) & (T0 > e)).long()
    return torch.where(base > 0, base + flows.long() * live, torch.zeros_like(base))


def sampler_kink_rays(z, weights, tau, T, near, far, u, eps=1e-3, zt=1e-4, rel=2e-4, cdf_tol=1e-5, path=None):
    """Rays on which the depth hypotheses' sampler (model/run_nerf_helpers.py:607-692 = run_nerf_helpers.py:340-445) stands within
    rounding of one of its SWITCHES: the cdf knot that decides a hypothesis' bin, the branch threshold |tau_r - tau_l| = zt, one
    of the max(eps, .) guards, or either end of the final clamp(t, eps, s_r - s_l).  Across a switch the sample is continuous
    (or hops a bin) but its GRADIENT is not -- zero on one side of a guard, the closed form's on the other -- so two correct
    evaluations return different gradients for the whole ray.  And t is the ill-conditioned quantity itself: in fp32 it moves by
    5e-4 of itself for a 3e-7 difference in T (third-seed campaign, seed 32 case 107: t = 0.99975e-3 on the path's T, 1.00025e-3
    on the oracle's, eps = 1e-3: clamped on one side, live on the other; seed 132 case 163: 1.0000227e-3 against 0.99999997e-3).
    Such rays are exempt from the upstream stage's bound, like rows with a ReLU unit within rounding of zero in the MLP tests; they
    are counted.  Two tests: (a) a switch quantity of the fp64 evaluation within `rel` of its threshold (or u within cdf_tol of a
    cdf knot); (b) with `path` = (tau, T, inds) of the path's own forward: the kernel's branch code evaluated in fp32 on those
    differs from the fp64 evaluation's, or the bins do.
    z, weights, tau, T, near, far, u: fp64 CPU tensors of ONE evaluation (the fp64 oracle's at the path's raw).  Returns bool [R]."""
    knots = torch.cat([near, z, far], dim=-1)
    cdf = torch.cat([torch.zeros_like(weights[..., :1]), torch.cumsum(weights, dim=-1)], dim=-1)
    cdf[:, -1] = 1.0
    inds = torch.searchsorted(cdf, u.contiguous(), right=True)
    below = torch.clamp(inds - 1, min=0)
    above = torch.clamp(inds, max=cdf.shape[-1] - 1)
    flag = ((u - torch.gather(cdf, -1, below)).abs() <= cdf_tol) | ((u - torch.gather(cdf, -1, above)).abs() <= cdf_tol)
    s0, s1 = torch.gather(knots, -1, below), torch.gather(knots, -1, above)
    T0 = torch.gather(T, -1, below)
    tau0, tau1 = torch.gather(tau, -1, below), torch.gather(tau, -1, above)
    dtau = tau[..., 1:] - tau[..., :-1]
    d = torch.gather(dtau, -1, torch.clamp(below, max=dtau.shape[-1] - 1))

    def near_(value, thr):
        thr = torch.as_tensor(thr, dtype=value.dtype)
        return (value - thr).abs() <= rel * torch.clamp(thr.abs(), min=1e-30)
    e = torch.full_like(T0, eps)
    q = (1 - u) / torch.max(e, T0)
    log_term = -torch.log(torch.max(e, q))
    span = torch.max(e, s1 - s0)
    rising, falling = d >= zt, d <= -zt
    live = rising | falling
    slope = torch.where(rising, tau1 - tau0, tau0 - tau1)
    disc = torch.where(rising, tau0 ** 2 + (2 * (tau1 - tau0) * log_term) / span, tau0 ** 2 - (2 * (tau0 - tau1) * log_term) / span)
    root = torch.sqrt(torch.max(e, disc))
    t = torch.where(rising, (s1 - s0) * (-tau0 + root), (s1 - s0) * (tau0 - root)) / torch.max(e, slope)
    flag |= near_(d.abs(), zt)
    flag |= live & (near_(T0, eps) | near_(q, eps) | near_(s1 - s0, eps) | near_(slope, eps) | near_(disc, eps) | near_(t, eps) |
                    near_(t, s1 - s0))
    if path is not None:
        tau_p, T_p, inds_p = (x.detach().cpu() for x in path)
        inds_p = inds_p.long()
        code64 = _sampler_branch_code(knots, tau, T, u, inds, eps, zt)
        code32 = _sampler_branch_code(knots.float(), tau_p.float(), T_p.float(), u.float(), inds_p, eps, zt)
        flag |= (code64 != code32) | (inds_p != inds)
    return flag.any(dim=-1)


def network_stage(query, sd, pts, viewdirs, g_raw_path, path_grads, tol, floor_frac):
    """query(sd, pts, viewdirs) -> raw: the oracle's network (fp dtype of its arguments).  sd: name -> fp32 tensor;
    g_raw_path: the cotangent the path's MLP backward received; path_grads: name -> the path's gradient (CPU).
    Returns (worst error, its tensor, list of violations): per tensor e = max |path - fp64| / max(own max, floor_frac x the
    network's largest entry); a tensor passes if e <= tol or e <= 3 x (the fp32 oracle's e against the same fp64)."""
    ref = {}
    for dt in (torch.float64, torch.float32):
        p = {k: v.detach().to(dt).clone().requires_grad_(True) for k, v in sd.items()}
        raw = query(p, pts.to(dt), viewdirs.to(dt))
        raw.backward(g_raw_path.detach().cpu().to(dt).reshape(raw.shape))
        ref[dt] = {k: (v.grad if v.grad is not None else torch.zeros_like(v)).double() for k, v in p.items()}
    g64, g32 = ref[torch.float64], ref[torch.float32]
    g_max = max(float(x.abs().max()) for x in g64.values())
    worst, which, bad = 0.0, None, []
    for name, g in path_grads.items():
        scale = max(float(g64[name].abs().max()), floor_frac * g_max, 1e-30)
        e = float((g.double().cpu() - g64[name]).abs().max()) / scale
        e32 = float((g32[name] - g64[name]).abs().max()) / scale
        if e > worst:
            worst, which = e, name
        if e > tol and e > 3.0 * e32:
            bad.append(f"{name}: {e:.2e} (fp32 oracle {e32:.2e})")
    return worst, which, bad
