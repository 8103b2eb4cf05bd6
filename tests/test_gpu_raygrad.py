"""Gradient with respect to the RAY BATCH (origins, directions, bounds, view directions) on a real MI355X.

The reference's render_rays is a chain of torch expressions, so autograd gives it d(loss)/d(ray_batch) for free
(run_plnerf.py:683-707 z_vals / pts, :516-550 + :604-617 interval lengths and depth map, :731-735 clamp / sort / fine
positions; run_nerf_helpers.py:24-54, 105-128 the network's inputs).  No reference training path asks for it -- the rays
are data -- but a drop-in must either give the same gradient or refuse, never a partial one.  Here:

  * plnerf_quad_bwd_rays (QuadratureFn): d / d z_vals, near, far, rays_d of raw2outputs, against the oracle's fp64 autograd
    on every output, both quadrature rules, ragged sample counts;
  * render_rays end to end, 64 + 128 samples, both networks: d / d ray_batch [R, 11] against the oracle's fp64 autograd on
    identical draws, in the exact fp32 mode and in the benchmarked f16x3;
  * plnerf_sample_pl_bwd_rays (SamplePlFn): d samples / d (z_vals, near, far), and with it the depth-supervised variant,
    whose depth hypotheses depend on the geometry through a sampler that is NOT detached (both quadrature rules: the
    piecewise-constant sampler's bins get theirs from SampleConstFn).

Tolerances are stated at each assertion, as fractions of the largest |g| of the column group.
"""
import functools

import pytest
import torch

from oracle import plnerf_oracle as orc
from test_gpu_parity import assert_close, g, make_net, quad_case

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def P():
    import plnerf_amd
    return plnerf_amd


@pytest.mark.parametrize("mode,cmode", [("linear", "midpoint"), ("linear", "left"), ("constant", "midpoint")])
@pytest.mark.parametrize("S", [64, 192, 37])
def test_quadrature_geometry_gradients_vs_oracle(P, mode, cmode, S):
    """d / d (z_vals, near, far, rays_d) of every raw2outputs output; the gradient of `raw` is the bits plnerf_quad_bwd
    gives without the request."""
    from plnerf_amd.functional import QuadratureFn
    R = 67
    raw, z, near, far, d, noise = quad_case(R, S, 11 + S)
    near = near - 0.25 * torch.rand(R, 1, generator=torch.Generator().manual_seed(S))      # (knots stay ordered)
    far = far + 0.25 * torch.rand(R, 1, generator=torch.Generator().manual_seed(S + 1))
    gen = torch.Generator().manual_seed(5)
    n = S + 1 if mode == "linear" else S
    cot = [torch.randn(R, 3, generator=gen), torch.randn(R, generator=gen), torch.randn(R, generator=gen),
           torch.randn(R, n, generator=gen) * 0.1, torch.randn(R, generator=gen)]
    for wb in (False, True):
        leaves_o = [t.double().clone().requires_grad_(True) for t in (raw, z, near, far, d)]
        ro = orc.raw2outputs(leaves_o[0], leaves_o[1], leaves_o[2], leaves_o[3], leaves_o[4], mode, cmode, white_bkgd=wb,
                             noise=noise.double())
        sum((a * b.double()).sum() for a, b in zip(ro[:5], cot)).backward()
        leaves_h = [g(t).clone().requires_grad_(True) for t in (raw, z, near, far, d)]
        rh = QuadratureFn.apply(*leaves_h, g(noise), mode, cmode, wb, False)
        sum((a * g(b)).sum() for a, b in zip(rh[:5], cot)).backward()
        for name, lh, lo in zip(("raw", "z_vals", "near", "far", "rays_d"), leaves_h, leaves_o):
            if mode == "constant" and name in ("near", "far"):      # (the classic rule does not use the bounds)
                assert lo.grad is None and float(lh.grad.abs().max()) == 0.0
                continue
            scale = float(lo.grad.abs().max())
            err = float((lh.grad.cpu().double() - lo.grad).abs().max()) / scale
            print(f"{mode}/{cmode}/S{S}/wb{wb} d/d{name}: {err:.2e} of max |g| = {scale:.3g}")
            # fp32 kernel against fp64 autograd: a few 1e-6 of the largest entry measured; 5e-5 asserted
            assert lh.grad.shape == lo.grad.shape and err <= 5e-5, (name, err)
        # the request does not change the gradient of `raw`
        raw_only = g(raw).clone().requires_grad_(True)
        r2 = QuadratureFn.apply(raw_only, g(z), g(near), g(far), g(d), g(noise), mode, cmode, wb, False)
        sum((a * g(b)).sum() for a, b in zip(r2[:5], cot)).backward()
        assert torch.equal(raw_only.grad, leaves_h[0].grad)


def test_sampler_bin_gradients_vs_oracle(P):
    """d samples / d (z_vals, near, far) of sample_pdf_reformulation (plnerf_sample_pl_bwd_rays) against autograd on the
    oracle: the closed-form inverse's dependence on the interval's ends (run_nerf_helpers.py:340-361), the flat-interval and
    NaN fall-backs to the left knot (:425, :432), the clamp's upper bound.  Yardstick and bound as in
    test_sampler_backward_vs_oracle_autograd: the fp64 oracle, twice the fp32 oracle's own distance from it + 1e-5 (the closed
    form cancels in fp32).  The gradients of tau and T are the bits plnerf_sample_pl_bwd gives without the request."""
    from plnerf_amd import functional as Fn

    def rel(a, b):
        return float((a.detach().cpu().double() - b).abs().max()) / (float(b.abs().max()) + 1e-300)

    for (R, S, N, seed) in [(64, 64, 64, 3), (33, 192, 128, 4), (16, 21, 77, 5)]:
        raw, z, near, far, d, _ = quad_case(R, S, seed)
        near = near - 0.25 * torch.rand(R, 1, generator=torch.Generator().manual_seed(seed))
        far = far + 0.25 * torch.rand(R, 1, generator=torch.Generator().manual_seed(seed + 1))
        gen = torch.Generator().manual_seed(seed)
        u = torch.rand(R, N, generator=gen) * 0.999
        cot = torch.randn(R, N, generator=gen)
        with torch.no_grad():
            _, _, _, w, _, tau, Tr = orc.raw2outputs(raw, z, near, far, d, "linear", "midpoint")
            tau[: R // 4, S // 3: S // 2] = tau[: R // 4, S // 3: S // 3 + 1]      # (flat stretches: the left-knot branch)

        def bin_grads(dt):
            leaves = [t.to(dt).clone().requires_grad_(True) for t in (z, near, far)]
            s_ref = orc.sample_pdf_reformulation(leaves[0], w.to(dt), tau.to(dt), Tr.to(dt), leaves[1], leaves[2], N, u=u.to(dt))[0]
            (s_ref * cot.to(dt)).sum().backward()
            return [l.grad for l in leaves]
        ref64, ref32 = bin_grads(torch.float64), bin_grads(torch.float32)
        leaves_h = [g(t).clone().requires_grad_(True) for t in (z, near, far)]
        tau_h, T_h = g(tau).requires_grad_(True), g(Tr).requires_grad_(True)
        s_hip = Fn.sample_pl(leaves_h[0], g(w), tau_h, T_h, leaves_h[1], leaves_h[2], g(u), 1e-4, 1e-3)
        (s_hip * g(cot)).sum().backward()
        for name, lh, b64, b32 in zip(("z_vals", "near", "far"), leaves_h, ref64, ref32):
            e_hip, e_orc = rel(lh.grad, b64), rel(b32, b64)
            print(f"sampler bins R={R} S={S} N={N}: d/d {name} err vs fp64 oracle: HIP {e_hip:.3e}, fp32 oracle {e_orc:.3e}")
            assert lh.grad.shape == b64.shape and e_hip <= 2 * e_orc + 1e-5, (name, e_hip, e_orc)
        tau_2, T_2 = g(tau).requires_grad_(True), g(Tr).requires_grad_(True)
        (Fn.sample_pl(g(z), g(w), tau_2, T_2, g(near), g(far), g(u), 1e-4, 1e-3) * g(cot)).sum().backward()
        assert torch.equal(tau_2.grad, tau_h.grad) and torch.equal(T_2.grad, T_h.grad)


def _query_fn(P):
    emb_fn, _ = P.get_embedder(10, 0)
    embd_fn, _ = P.get_embedder(4, 0)
    return lambda inputs, viewdirs, fn: P.run_network(inputs, viewdirs, fn, emb_fn, embd_fn)


def _preactivations(sd64, pts64, vd64):
    """Every ReLU unit's pre-activation in fp64, [rows, 2176] (pts64 [R, S, 3], vd64 [R, 3]); also (sigma, rgb) rows."""
    F = torch.nn.functional
    R, S = pts64.shape[:2]
    enc_xyz = orc.positional_encoding(pts64.reshape(-1, 3), orc.XYZ_FREQS)
    enc_dir = orc.positional_encoding(vd64[:, None, :].expand(R, S, 3).reshape(-1, 3), orc.DIR_FREQS)
    h, zs = enc_xyz, []
    for i in range(orc.DEPTH):
        zs.append(F.linear(h, sd64[f"pts_linears.{i}.weight"], sd64[f"pts_linears.{i}.bias"]))
        h = F.relu(zs[-1])
        if i == orc.SKIP_AFTER:
            h = torch.cat([enc_xyz, h], -1)
    sigma = F.linear(h, sd64["alpha_linear.weight"], sd64["alpha_linear.bias"])
    feat = F.linear(h, sd64["feature_linear.weight"], sd64["feature_linear.bias"])
    zs.append(F.linear(torch.cat([feat, enc_dir], -1), sd64["views_linears.0.weight"], sd64["views_linears.0.bias"]))
    rgb = F.linear(F.relu(zs[-1]), sd64["rgb_linear.weight"], sd64["rgb_linear.bias"])
    return torch.cat(zs, -1), sigma, rgb


@functools.lru_cache(maxsize=None)
def _decisive_state_dict(seed, box=4.5, n_probe=20000, margin=1.5, max_band=None):
    """The closed-form weights with every hidden unit's bias moved so that the unit is decisively ON or decisively OFF
    over the whole scene box (at random, half each), and the heads rescaled to a scene-like density / colour range.

    Why: a ray's gradient sums the input gradients of its 2 Ns + Ni rows through ~2,200 ReLU units each, and with ordinary
    weights 5 % of ROWS hold a unit within fp32 rounding of zero, which takes the other side of its ReLU in another
    arithmetic and moves the gradient by its whole contribution -- nearly every ray would hold one.
    With these weights the network is affine in its ENCODED inputs on the data (asserted: the smallest |pre-activation| over
    the test's own rows), so the comparison tests what is new here at full strictness: the positions / depths / clamp / sort
    chain, the quadrature's geometry gradient and the encoding's derivative behind plnerf_mlp_input_grad, composed.
    max_band: see below -- without the high bands fp32's rounding of a sample POSITION (1e-6) no longer reaches the gradient
    multiplied by the top frequency 2^9 (d/dx sin(512 x) moves by 512 * 1e-6 of itself), and fp32 meets fp64 at 1e-5."""
    F = torch.nn.functional
    sd = {k: v.double().clone() for k, v in orc.closed_form_state_dict(seed, True).items()}
    if max_band is not None:      # a network that ignores the position encoding's bands k >= max_band (sin / cos of 2^k x)
        sd["pts_linears.0.weight"][:, 3 + 6 * max_band:orc.XYZ_CH] = 0.0
        sd[f"pts_linears.{orc.SKIP_AFTER + 1}.weight"][:, 3 + 6 * max_band:orc.XYZ_CH] = 0.0
    gen = torch.Generator().manual_seed(1000 + seed)
    pts = (torch.rand(n_probe, 3, generator=gen, dtype=torch.float64) * 2 - 1) * box
    vd = F.normalize(torch.randn(n_probe, 3, generator=gen, dtype=torch.float64), dim=-1)
    enc_xyz, enc_dir = orc.positional_encoding(pts, orc.XYZ_FREQS), orc.positional_encoding(vd, orc.DIR_FREQS)

    def decide(z, key):      # move every unit's range over the probe points away from zero, on or off at random
        lo, hi = z.min(0).values, z.max(0).values
        sign = torch.where(torch.rand(z.shape[1], generator=gen) < 0.5, 1.0, -1.0).double()
        shift = -0.5 * (hi + lo) + sign * (margin * 0.5 * (hi - lo) + 0.05)
        sd[key] += shift
        return z + shift
    h = enc_xyz
    for i in range(orc.DEPTH):      # (one walk: a layer's shift is known before the next layer is evaluated)
        h = F.relu(decide(F.linear(h, sd[f"pts_linears.{i}.weight"], sd[f"pts_linears.{i}.bias"]), f"pts_linears.{i}.bias"))
        if i == orc.SKIP_AFTER:
            h = torch.cat([enc_xyz, h], -1)
    sigma = F.linear(h, sd["alpha_linear.weight"], sd["alpha_linear.bias"])
    feat = F.linear(h, sd["feature_linear.weight"], sd["feature_linear.bias"])
    hv = F.relu(decide(F.linear(torch.cat([feat, enc_dir], -1), sd["views_linears.0.weight"], sd["views_linears.0.bias"]),
                       "views_linears.0.bias"))
    rgb = F.linear(hv, sd["rgb_linear.weight"], sd["rgb_linear.bias"])
    # density ~ N(0.3, 0.06^2): positive everywhere (its own ReLU is decisive too), rays end semi-transparent.  Empty space
    # would do more than flip that ReLU: between two knots of equal density (tau = 0 on both sides) the sampler returns the
    # LEFT KNOT itself (run_nerf_helpers.py:425, |d tau| < zero_threshold) -- a tie in the sort of run_plnerf.py:734 for every
    # sample that lands in empty space (_both_gradients deals with the few ties that remain).
    k_s = 0.06 / float(sigma.std())
    sd["alpha_linear.weight"] *= k_s
    sd["alpha_linear.bias"] = (sd["alpha_linear.bias"] - sigma.mean()) * k_s + 0.3
    k_c = 1.5 / rgb.std(0)
    sd["rgb_linear.weight"] *= k_c[:, None]
    sd["rgb_linear.bias"] = (sd["rgb_linear.bias"] - rgb.mean(0)) * k_c
    return {k: v.float() for k, v in sd.items()}


GROUPS = (("rays_o", slice(0, 3)), ("rays_d", slice(3, 6)), ("near", slice(6, 7)), ("far", slice(7, 8)),
          ("viewdirs", slice(8, 11)))
KEYS = ("rgb_map", "depth_map", "acc_map", "disp_map", "rgb0", "depth0", "acc0", "z_std")


def _both_gradients(P, batch, sd_c, sd_f, precision, mode, Ns, Ni, cot, **extra):
    """d (sum of cot . maps) / d ray_batch: (oracle fp64, HIP path, oracle internals, the two HIP networks, the maps' keys).

    The fine pass of the fp64 side runs on the HIP path's OWN importance samples in the HIP path's OWN sort order (taken
    from render.STAGE_TAP).  The samples are detached on both sides (run_plnerf.py:728), so this changes no gradient path;
    what it removes is an ambiguity of the reference itself: where the density is flat between two knots the sampler returns
    the left knot itself (run_nerf_helpers.py:425) -- several samples per ray exactly ON a coarse depth -- the sort of :734
    then holds ties, torch.sort is not stable, and the two
    tied slots hand different gradients to near / far depending on which of the equal values landed where (measured
    before this: near / far off by up to 5e-2 on exactly the rays with ties, every other column at 7e-6)."""
    import sys
    render_mod = sys.modules["plnerf_amd.render"]
    kw = dict(perturb=1.0, N_importance=Ni, white_bkgd=True, pytest=True, **extra)
    sd_c64 = {k: v.double() for k, v in sd_c.items()}
    sd_f64 = {k: v.double() for k, v in sd_f.items()}
    net_c, net_f = make_net(P, sd_c, precision), make_net(P, sd_f, precision)
    b_h = g(batch).clone().requires_grad_(True)
    tap = {}
    render_mod.STAGE_TAP = tap
    try:
        got = P.render_rays(b_h, net_c, _query_fn(P), Ns, mode, "midpoint", network_fine=net_f, **kw)
    finally:
        render_mod.STAGE_TAP = None
    sum((got[k] * g(cot[k])).sum() for k in KEYS).backward()
    assert b_h.grad is not None and b_h.grad.shape == batch.shape and bool(torch.isfinite(b_h.grad).all())

    b_o = batch.double().clone().requires_grad_(True)
    ref, internals = orc.render_rays(b_o, sd_c64, sd_f64, Ns, mode, "midpoint", return_internals=True, **kw)
    near, far = b_o[:, 6:7], b_o[:, 7:8]
    z_new = torch.clamp(tap["z_samples"].detach().cpu().double(), near, far)
    z_fine = torch.gather(torch.cat([internals["z_coarse"], z_new], -1), -1, tap["sort_order"].cpu())
    assert float((z_fine.detach() - tap["z_fine"].detach().cpu().double()).abs().max()) <= 1e-5      # (the same depths)
    fine = orc.fine_stage(b_o, sd_f64, z_fine, mode, "midpoint", white_bkgd=True)
    ref = dict(ref, z_std=torch.std(z_new, dim=-1, unbiased=False), **{k: fine[k] for k in KEYS[:4]})
    # (the forward is the usual one; against FP64 it carries fp32's rounding of a sample position times the encoding's top
    # frequency, 2^9: ~1e-4 on a map -- the 1e-5 contract is against the reference's own fp32, test_gpu_parity.py)
    for k in ("rgb_map", "acc_map", "rgb0", "depth0", "acc0"):
        assert float((got[k].detach().cpu().double() - ref[k].detach()).abs().max()) <= 5e-4, k
    sum((ref[k] * cot[k].double()).sum() for k in KEYS).backward()
    return b_o.grad, b_h.grad.cpu().double(), internals, (net_c, net_f), list(KEYS)


def _cotangents(R, seed=17):
    gen = torch.Generator().manual_seed(seed)
    return {k: torch.randn(R, 3 if k.startswith("rgb") else 1, generator=gen).squeeze(-1) for k in KEYS}


@pytest.mark.parametrize("precision,mode,max_band,tol", [
    ("fp32", "linear", 3, 2e-5), ("fp32", "constant", 3, 2e-5),      # the composition, strictly (measured 4e-6 / 6e-6)
    ("fp32", "linear", None, 1e-3), ("f16x3", "linear", None, 4e-3), ("f16x3", "linear", 3, 1.5e-3)])
def test_render_rays_gradient_wrt_the_ray_batch_vs_oracle(P, precision, mode, max_band, tol):
    """64 + 128 samples, both networks, jittered depths and sampler draws replayed on both sides (pytest=True), white
    background: the gradient of a random functional of every returned map with respect to ray_batch [R, 11] against the
    oracle in fp64, on EVERY ray -- with networks whose ReLU units are decisively on or off (_decisive_state_dict) and
    the fine pass on identical samples in identical order (_both_gradients).

    Tolerances, per ray, as a fraction of (the ray's own largest entry + the column group's median magnitude):
      * networks that read the encoding's bands below 2^3 only: fp32 kernels 2e-5 (measured 4e-6 on the worst ray of the
        worst column) -- the strict test of what is new: a missing or mis-scaled term of the chain shows at 1e-2 or
        more; f16x3 1.5e-3 (measured 3.7e-4: half planes in the backward);
      * all ten bands: fp32 1e-3 (measured 3.3e-4), f16x3 4e-3 (measured 1.3e-3) -- fp32's own rounding of a coarse depth
        (1e-7 of 4) moves d/dx sin(2^9 x) by 2^9 times that per row before any kernel is involved, and rows cancel
        within a ray."""
    R, Ns, Ni = 48, 64, 128
    batch, _ = orc.synthetic_blender_rays(R, seed=5)
    sd_c, sd_f = _decisive_state_dict(0, max_band=max_band), _decisive_state_dict(1, max_band=max_band)
    cot = _cotangents(R)
    ref_g, got_g, internals, nets, keys = _both_gradients(P, batch, sd_c, sd_f, precision, mode, Ns, Ni, cot)
    with torch.no_grad():      # the premise: no unit of either network is near zero on any row of this test
        o64, d64, vd64 = batch[:, 0:3].double(), batch[:, 3:6].double(), batch[:, 8:11].double()
        for sd, z in ((sd_c, internals["z_coarse"]), (sd_f, internals["z_fine"])):
            zs, sigma, _ = _preactivations({k: v.double() for k, v in sd.items()}, o64[:, None, :] + d64[:, None, :] * z.detach()[..., None], vd64)
            assert float(zs.abs().min()) > 1e-3, float(zs.abs().min())
            assert float(sigma.min()) > 0.02      # (no empty space: see _decisive_state_dict)
    # A column group's gradients span orders of magnitude from ray to ray (the far bound's: 1e10 where the last sample
    # sits on it), so each ray is held to `tol` of ITS OWN largest entry plus the group's median magnitude.
    worst = {}
    for name, cols in GROUPS:
        mag = ref_g[:, cols].abs().max(-1).values
        floor = float(mag.median())
        ray_err = (got_g[:, cols] - ref_g[:, cols]).abs().max(-1).values / (mag + floor)
        print(f"{precision}/{mode}/bands<{max_band} d/d {name}: |g| per ray median {floor:.3g}, max {float(mag.max()):.3g}; error / "
              f"(own |g| + median): median ray {float(ray_err.median()):.2e}, worst {float(ray_err.max()):.2e}")
        assert floor > 0.0
        worst[name] = float(ray_err.max())
    assert max(worst.values()) <= tol, worst
    # the parameters' gradients are those of the same call on a batch that does not ask
    net_c, net_f = nets
    with_rays = [p.grad.clone() for p in list(net_c.parameters()) + list(net_f.parameters())]
    net_c.zero_grad(); net_f.zero_grad()
    got2 = P.render_rays(g(batch), net_c, _query_fn(P), Ns, mode, "midpoint", network_fine=net_f, perturb=1.0,
                         N_importance=Ni, white_bkgd=True, pytest=True)
    sum((got2[k] * g(cot[k])).sum() for k in keys).backward()
    for a, p in zip(with_rays, list(net_c.parameters()) + list(net_f.parameters())):
        assert float((a - p.grad).abs().max()) <= 1e-5 * float(p.grad.abs().max()) + 1e-12


def test_render_rays_gradient_wrt_the_ray_batch_with_ordinary_weights(P):
    """The same comparison with the closed-form weights every other test uses: nearly every ray now holds a unit within fp32
    rounding of zero (5 % of rows do, test_input_gradients_match_the_oracle shows the mechanism row by row), so the bound
    is statistical: the median ray at the fp32 tolerance, no ray beyond 5e-2 of the largest gradient."""
    R, Ns, Ni = 48, 64, 128
    batch, _ = orc.synthetic_blender_rays(R, seed=5)
    ref_g, got_g, _, _, _ = _both_gradients(P, batch, orc.closed_form_state_dict(0, True), orc.closed_form_state_dict(1, True),
                                            "fp32", "linear", Ns, Ni, _cotangents(R))
    for name, cols in GROUPS:
        mag = ref_g[:, cols].abs().max(-1).values
        ray_err = (got_g[:, cols] - ref_g[:, cols]).abs().max(-1).values / (mag + float(mag.median()))
        print(f"ordinary weights d/d {name}: median ray {float(ray_err.median()):.2e}, worst {float(ray_err.max()):.2e}, "
              f"{int((ray_err > 1e-4).sum())}/{R} rays beyond 1e-4")
        assert float(ray_err.median()) <= 5e-4 and float(ray_err.max()) <= 5e-2, (name, float(ray_err.max()))


def test_single_pass_lindisp_and_one_column_group(P):
    """N_importance = 0 (one network, depths linear in disparity, `left` colours) and a gradient asked for the origins only."""
    R, Ns = 40, 48
    batch, _ = orc.synthetic_blender_rays(R, seed=9)
    sd = _decisive_state_dict(2)
    cot = torch.randn(R, 3, generator=torch.Generator().manual_seed(3))
    b_o = batch.double().clone().requires_grad_(True)
    ref = orc.render_rays(b_o, {k: v.double() for k, v in sd.items()}, None, Ns, "linear", "left", lindisp=True)
    ((ref["rgb_map"] * cot.double()).sum() + ref["depth_map"].sum()).backward()
    net = make_net(P, sd, "fp32")
    o = g(batch[:, 0:3]).clone().requires_grad_(True)
    b_h = torch.cat([o, g(batch[:, 3:])], -1)
    got = P.render_rays(b_h, net, _query_fn(P), Ns, "linear", "left", lindisp=True)
    ((got["rgb_map"] * g(cot)).sum() + got["depth_map"].sum()).backward()
    mag = b_o.grad[:, 0:3].abs().max(-1).values
    err = (o.grad.cpu().double() - b_o.grad[:, 0:3]).abs().max(-1).values / (mag + float(mag.median()))
    print(f"single pass, lindisp: d/d rays_o median {float(err.median()):.2e}, worst {float(err.max()):.2e} of (own |g| + median "
          f"{float(mag.median()):.3g})")
    assert float(err.max()) <= 5e-4


def test_camera_pose_gradient_through_render(P):
    """The use a ray-batch gradient has: d loss / d c2w through render() (run_plnerf.py:110-175: get_rays, unit view
    directions, packing, chunks of render_rays) -- every host step of the mirror is a torch expression and stays on the
    tape.  6 x 8 pixels, 32 + 48 samples, two chunks; against the same chain in fp64."""
    H, W, Ns, Ni = 6, 8, 32, 48
    K = [[10.0, 0.0, W / 2], [0.0, 10.0, H / 2], [0.0, 0.0, 1.0]]
    c2w = orc.pose_spherical(30.0, -30.0, 4.0)[:3, :4]
    sd_c, sd_f = _decisive_state_dict(0, max_band=3), _decisive_state_dict(1, max_band=3)
    gen = torch.Generator().manual_seed(23)
    cot_rgb, cot_depth = torch.randn(H, W, 3, generator=gen), torch.randn(H, W, generator=gen)
    kw = dict(N_samples=Ns, mode="linear", color_mode="midpoint", perturb=0.0, N_importance=Ni, white_bkgd=True)
    # fp64: the same chain written out
    c_o = c2w.double().clone().requires_grad_(True)
    px = torch.arange(W, dtype=torch.float64)[None, :].expand(H, W)
    py = torch.arange(H, dtype=torch.float64)[:, None].expand(H, W)
    cam = torch.stack([(px - K[0][2]) / K[0][0], -(py - K[1][2]) / K[1][1], -torch.ones_like(px)], -1)
    d = torch.sum(cam[..., None, :] * c_o[:3, :3], -1).reshape(-1, 3)
    o = c_o[:3, -1].expand(d.shape)
    ones = torch.ones_like(d[:, :1])
    batch = torch.cat([o, d, 2.0 * ones, 6.0 * ones, d / torch.norm(d, dim=-1, keepdim=True)], -1)
    ref = orc.render_rays(batch, {k: v.double() for k, v in sd_c.items()}, {k: v.double() for k, v in sd_f.items()}, Ns,
                          "linear", "midpoint", perturb=0.0, N_importance=Ni, white_bkgd=True)
    ((ref["rgb_map"].reshape(H, W, 3) * cot_rgb.double()).sum() + (ref["depth_map"].reshape(H, W) * cot_depth.double()).sum()).backward()
    # the mirror
    net_c, net_f = make_net(P, sd_c, "fp32"), make_net(P, sd_f, "fp32")
    c_h = g(c2w).clone().requires_grad_(True)
    rgb, disp, acc, extras = P.render(H, W, K, chunk=32, c2w=c_h, ndc=False, near=2.0, far=6.0, use_viewdirs=True,
                                      network_fn=net_c, network_fine=net_f, network_query_fn=_query_fn(P), **kw)
    assert rgb.shape == (H, W, 3) and float((rgb.detach().cpu().double() - ref["rgb_map"].detach().reshape(H, W, 3)).abs().max()) <= 5e-4
    ((rgb * g(cot_rgb)).sum() + (extras["depth_map"] * g(cot_depth)).sum()).backward()
    scale = float(c_o.grad.abs().max())
    err = float((c_h.grad.cpu().double() - c_o.grad).abs().max()) / scale
    print(f"d loss / d c2w: max |g| {scale:.3g}, largest error {err:.2e} of it")
    assert c_h.grad.shape == (3, 4) and err <= 2e-5


@pytest.mark.parametrize("mode,Ni", [("linear", 0), ("linear", 40), ("constant", 40)])
def test_depth_variant_ray_batch_gradient_vs_oracle(P, mode, Ni):
    """The depth-supervised render_rays (run_nerf_sample_based_depth.py:792-958) on a ray batch that requires a gradient:
    its depth hypotheses stay attached to the sampler, so d pred_hyp / d (bins) joins the chain (plnerf_sample_pl_bwd_rays).
    A random functional of rgb_map, depth_map, pred_hyp (and rgb0) against the oracle on shared draws.

    The loss runs through the sampler's ill-conditioned closed form and through ReLUs near zero, where the reference's own
    fp32 autograd sits ~1e-2 of max |g| from fp64 (test_depth_variant_gradients_vs_oracle): yardstick = the fp64 oracle,
    bound = twice the fp32 oracle's own distance from it + 1e-3, per column group.  With a fine pass (Ni > 0) near / far are
    left out: their gradient depends on how the sort of :906 orders a sample that ties with a coarse depth (_both_gradients
    explains; the single-pass case has no merge and checks them)."""
    import sys
    from plnerf_amd import depth as Dp
    from test_gpu_parity import _depth_setup
    R, Ns = 24, 32
    Dp, kw, _, _ = _depth_setup(P, {"N_importance": 40, "N_samples": Ns, "space_carving_weight": 0.05})      # (both networks exist)
    # (piecewise-constant mode on default-initialised networks, as test_depth_variant_constant_mode_vs_oracle: with the
    # "sharpened" ones most bins are empty and sample_pdf divides by cdf steps of ~1e-5 -- fp32 noise, the reference's too)
    sharp = mode == "linear"
    if not sharp:
        kw["network_fn"].load_state_dict(orc.closed_form_state_dict_depth(0, False))
        kw["network_fine"].load_state_dict(orc.closed_form_state_dict_depth(1, False))
    batch, _ = orc.synthetic_blender_rays(R, seed=13)
    gen = torch.Generator().manual_seed(13)
    n_hyp = Ni if Ni > 0 else Ns
    t_rand, u_fine, u_hyp = torch.rand(R, Ns, generator=gen), torch.rand(R, max(Ni, 1), generator=gen) * 0.999, \
        torch.rand(R, n_hyp, generator=gen) * 0.999
    cot = {"rgb_map": torch.randn(R, 3, generator=gen), "depth_map": torch.randn(R, generator=gen),
           "pred_hyp": torch.randn(R, n_hyp, generator=gen) * 0.2, "rgb0": torch.randn(R, 3, generator=gen)}
    keys = [k for k in cot if Ni > 0 or k != "rgb0"]

    cur = {}

    def oracle(dt):
        cur["dt"] = dt
        sd_c = {k: v.to(dt) for k, v in orc.closed_form_state_dict_depth(0, sharp).items()}
        sd_f = {k: v.to(dt) for k, v in orc.closed_form_state_dict_depth(1, sharp).items()}
        b = batch.to(dt).clone().requires_grad_(True)
        ret = orc.render_rays_depth(b, sd_c, sd_f, Ns, mode, "midpoint", perturb=1.0, N_importance=Ni, white_bkgd=True,
                                    t_rand=t_rand.to(dt), u_fine=u_fine.to(dt), cached_u=u_hyp.to(dt) if Ni > 0 else None)
        sum((ret[k] * cot[k].to(dt)).sum() for k in keys).backward()
        return b.grad.double(), ret
    if Ni == 0:      # (the single pass draws its hypotheses' u itself: replay the oracle's)
        orig_u = orc._draw_u_depth
        orc._draw_u_depth = lambda R_, n, det, pyt, load_u: u_hyp.to(cur["dt"]) if load_u is None else load_u
    try:
        g64, _ = oracle(torch.float64)
        g32, ret32 = oracle(torch.float32)
    finally:
        if Ni == 0:
            orc._draw_u_depth = orig_u
    dmod = sys.modules[Dp.__name__]
    rmod = sys.modules[Dp.__name__.rsplit(".", 1)[0] + ".render"]
    orig = dmod._draw_t_rand, rmod._draw_u, dmod._draw_u
    b_h = g(batch).clone().requires_grad_(True)
    try:
        dmod._draw_t_rand = lambda *a, **k: g(t_rand)
        rmod._draw_u = lambda *a, **k: g(u_fine)
        if Ni == 0:
            dmod._draw_u = lambda *a, **k: g(u_hyp)
        ret = Dp.render_rays(b_h, cached_u=g(u_hyp) if Ni > 0 else None, **dict(kw, N_importance=Ni, mode=mode))
    finally:
        dmod._draw_t_rand, rmod._draw_u, dmod._draw_u = orig
    assert float((ret["pred_hyp"].detach().cpu() - ret32["pred_hyp"].detach()).abs().max()) <= 5e-3      # (same hypotheses)
    sum((ret[k] * g(cot[k])).sum() for k in keys).backward()
    assert b_h.grad is not None and bool(torch.isfinite(b_h.grad).all())
    got = b_h.grad.cpu().double()
    for name, cols in GROUPS:
        if Ni > 0 and name in ("near", "far"):
            continue
        scale = float(g64[:, cols].abs().max())
        e_hip = float((got[:, cols] - g64[:, cols]).abs().max()) / scale
        e_orc = float((g32[:, cols] - g64[:, cols]).abs().max()) / scale
        e_32 = float((got[:, cols] - g32[:, cols]).abs().max()) / scale      # (against the reference's own arithmetic)
        print(f"depth variant {mode} Ni={Ni} d/d {name}: max |g| {scale:.3g}; vs fp64 oracle: HIP {e_hip:.2e}, fp32 oracle {e_orc:.2e}; "
              f"HIP vs fp32 oracle {e_32:.2e}")
        assert scale > 0.0 and e_hip <= 2 * e_orc + 1e-3, (name, e_hip, e_orc)
        assert e_32 <= 0.5 * e_orc + 1e-3, (name, e_32, e_orc)


def test_sample_pdf_bin_gradients_vs_oracle(P):
    """d samples / d bins of sample_pdf (run_nerf_helpers.py:241-284: samples = b0 + t (b1 - b0)) -- SampleConstFn -- against
    the oracle's autograd in fp64; the gradient of the weights is the bits plnerf_sample_const_bwd gives without the request."""
    from plnerf_amd import functional as Fn
    for (R, B, N, seed) in [(64, 63, 64, 3), (33, 191, 128, 4), (16, 20, 77, 5)]:
        gen = torch.Generator().manual_seed(seed)
        bins = torch.sort(2.0 + 4.0 * torch.rand(R, B, generator=gen), -1).values
        w = torch.rand(R, B - 1, generator=gen) ** 3
        u = torch.rand(R, N, generator=gen) * 0.999
        cot = torch.randn(R, N, generator=gen)
        def bin_grad(dt):
            b_o = bins.to(dt).clone().requires_grad_(True)
            (orc.sample_pdf(b_o, w.to(dt), N, u=u.to(dt)) * cot.to(dt)).sum().backward()
            return b_o.grad.double()
        ref64, ref32 = bin_grad(torch.float64), bin_grad(torch.float32)
        b_h, w_h = g(bins).clone().requires_grad_(True), g(w).clone().requires_grad_(True)
        (Fn.sample_const(b_h, w_h, g(u)) * g(cot)).sum().backward()
        scale = float(ref64.abs().max())
        err = float((b_h.grad.cpu().double() - ref64).abs().max()) / scale
        e_orc = float((ref32 - ref64).abs().max()) / scale
        print(f"sample_pdf bins R={R} B={B} N={N}: d/d bins vs fp64 oracle: HIP {err:.2e}, fp32 oracle {e_orc:.2e} of max |g|")
        # (t = (u - cdf) / (cdf step) cancels in fp32: the yardstick is the fp32 oracle's own distance from fp64)
        assert b_h.grad.shape == bins.shape and err <= 2 * e_orc + 1e-5
        w_2 = g(w).clone().requires_grad_(True)
        (Fn.sample_const(g(bins), w_2, g(u)) * g(cot)).sum().backward()
        assert torch.equal(w_2.grad, w_h.grad)
