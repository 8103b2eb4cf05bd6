"""CPU oracle for the PL-NeRF volume-rendering hot path.  TEST INFRASTRUCTURE ONLY.

This file is a from-the-maths restatement, in plain fp32 PyTorch on the CPU, of
the reference path named by BASELINE.json (mikacuy/PL-NeRF: run_plnerf.py and
run_nerf_helpers.py).  Each function cites the reference file:line it follows.
It exists to CHECK the HIP path (tests/, __graft_entry__.smoke()) and to be
TIMED as bench.py's `cpu_baseline` leg (kind "port").  Nothing in the product
package (pl-nerf_amd/) may import it; the product fails loudly without the HIP
library instead of falling back to this code.

Parity pin: the reference ships no tests or golden vectors,
so this oracle is pinned against outputs of the reference itself, generated in
the build container by tests/golden/make_golden.py (which imports
/root/reference read-only) and committed as tests/golden/*.npz;
tests/test_oracle_golden.py re-checks the oracle against them on every run.

Floating point: everything is fp32 like the reference; `searchsorted` indices
are int64.  torch CPU internals the fixtures depend on (fp64-accumulated
cumsum/cumprod, the vectorised fp32 `sum` tree, two-sided `linspace`) are used
here through the same torch ops, so the oracle reproduces them by construction.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

# ----------------------------------------------------------------------------
# Network description (run_nerf_helpers.py:76-104, defaults run_plnerf.py:784-825)
# ----------------------------------------------------------------------------
XYZ_FREQS = 10          # multires
DIR_FREQS = 4           # multires_views
XYZ_CH = 3 + 6 * XYZ_FREQS   # 63
DIR_CH = 3 + 6 * DIR_FREQS   # 27
WIDTH = 256
DEPTH = 8
SKIP_AFTER = 4          # skips=[4]: input re-injected after layer 4's ReLU


def param_shapes():
    """state_dict keys and shapes in nn.Module registration order
    (run_nerf_helpers.py:87-101): trunk, view layer, feature, alpha, rgb."""
    shapes = []
    for i in range(DEPTH):
        fan_in = XYZ_CH if i == 0 else (WIDTH + XYZ_CH if i == SKIP_AFTER + 1 else WIDTH)
        shapes.append((f"pts_linears.{i}.weight", (WIDTH, fan_in)))
        shapes.append((f"pts_linears.{i}.bias", (WIDTH,)))
    shapes.append(("views_linears.0.weight", (WIDTH // 2, WIDTH + DIR_CH)))
    shapes.append(("views_linears.0.bias", (WIDTH // 2,)))
    shapes.append(("feature_linear.weight", (WIDTH, WIDTH)))
    shapes.append(("feature_linear.bias", (WIDTH,)))
    shapes.append(("alpha_linear.weight", (1, WIDTH)))
    shapes.append(("alpha_linear.bias", (1,)))
    shapes.append(("rgb_linear.weight", (3, WIDTH // 2)))
    shapes.append(("rgb_linear.bias", (3,)))
    return shapes


def closed_form_state_dict(seed=0, sharpen=False):
    """Deterministic, RNG-free network weights (so fixtures need not store the
    4.8 MB of parameters): a sine hash scaled to nn.Linear's default
    U(-1/sqrt(fan_in), 1/sqrt(fan_in)) range.  `sharpen` scales the density head
    so rays saturate (acc ~ 1) and every sampler branch is exercised
    (SURVEY.md section 8d)."""
    sd = {}
    for li, (name, shape) in enumerate(param_shapes()):
        fan_in = shape[1] if len(shape) == 2 else None
        if fan_in is None:
            wname = name.replace("bias", "weight")
            fan_in = dict(param_shapes())[wname][1]
        bound = 1.0 / math.sqrt(fan_in)
        n = int(np.prod(shape))
        idx = np.arange(n, dtype=np.float64)
        phase = 12.9898 * (idx + 1.0) + 78.233 * (li + 1.0) + 37.719 * (seed + 1.0)
        h = np.sin(phase) * 43758.5453
        frac = h - np.floor(h)                       # in [0,1)
        w = ((2.0 * frac - 1.0) * bound).astype(np.float32).reshape(shape)
        sd[name] = torch.from_numpy(w)
    if sharpen:
        sd["alpha_linear.weight"] = sd["alpha_linear.weight"] * 30.0
        sd["alpha_linear.bias"] = sd["alpha_linear.bias"] + 0.5
    return sd


# ----------------------------------------------------------------------------
# a3: positional encoding (run_nerf_helpers.py:24-72)
# ----------------------------------------------------------------------------
def positional_encoding(x, n_freqs):
    """gamma(x) = [x, sin(2^0 x), cos(2^0 x), ..., sin(2^(L-1) x), cos(2^(L-1) x)];
    every block is as wide as x (run_nerf_helpers.py:31-54).  The frequencies are
    exact powers of two (line 41)."""
    blocks = [x]
    for k in range(n_freqs):
        xs = x * float(2 ** k)
        blocks.append(torch.sin(xs))
        blocks.append(torch.cos(xs))
    return torch.cat(blocks, dim=-1)


# ----------------------------------------------------------------------------
# a4: the MLP (run_nerf_helpers.py:105-128)
# ----------------------------------------------------------------------------
def nerf_mlp(sd, embedded, return_hidden=False):
    """embedded: [N, 63+27] = gamma(xyz) ++ gamma(viewdir).  Returns [N,4] =
    (r,g,b,sigma) pre-activation.  Layer 5 consumes cat([gamma(xyz), h4]) with the
    63 encoding channels FIRST (lines 111-112)."""
    enc_xyz, enc_dir = embedded[..., :XYZ_CH], embedded[..., XYZ_CH:]
    h = enc_xyz
    hidden = []
    for i in range(DEPTH):
        h = F.relu(F.linear(h, sd[f"pts_linears.{i}.weight"], sd[f"pts_linears.{i}.bias"]))
        hidden.append(h)
        if i == SKIP_AFTER:
            h = torch.cat([enc_xyz, h], dim=-1)
    sigma = F.linear(h, sd["alpha_linear.weight"], sd["alpha_linear.bias"])       # line 115
    feat = F.linear(h, sd["feature_linear.weight"], sd["feature_linear.bias"])    # line 116
    hv = F.relu(F.linear(torch.cat([feat, enc_dir], dim=-1),
                         sd["views_linears.0.weight"], sd["views_linears.0.bias"]))  # 117-121
    rgb = F.linear(hv, sd["rgb_linear.weight"], sd["rgb_linear.bias"])            # line 123
    out = torch.cat([rgb, sigma], dim=-1)                                         # line 124
    if return_hidden:
        return out, hidden + [feat, hv]
    return out


def query_network(sd, pts, viewdirs, netchunk=65536):
    """a2: run_plnerf.py:78-92.  pts [R,S,3], viewdirs [R,3] -> raw [R,S,4].  The
    view direction is broadcast to every sample and then encoded (85-88)."""
    R, S = pts.shape[0], pts.shape[1]
    flat = pts.reshape(-1, 3)
    emb = positional_encoding(flat, XYZ_FREQS)
    dirs = viewdirs[:, None, :].expand(R, S, 3).reshape(-1, 3)
    emb = torch.cat([emb, positional_encoding(dirs, DIR_FREQS)], dim=-1)
    outs = [nerf_mlp(sd, emb[i:i + netchunk]) for i in range(0, emb.shape[0], netchunk)]
    return torch.cat(outs, 0).reshape(R, S, 4)


# ----------------------------------------------------------------------------
# a5 / a6: interval opacities and transmittance
# ----------------------------------------------------------------------------
def weights_piecewise_linear(raw, z, near, far, rays_d, noise=0.0):
    """run_plnerf.py:516-550.  Returns (weights [R,S+1], tau [R,S+2], T [R,S+2]).
    Knots are [near, z, far]; tau = relu([1e-10, sigma+noise, 1e10]); each
    interval's opacity integrates the trapezoid of tau over its length times |d|."""
    R = raw.shape[0]
    knots = torch.cat([near, z, far], dim=-1)
    seg = (knots[..., 1:] - knots[..., :-1]) * torch.norm(rays_d[..., None, :], dim=-1)
    lo = torch.full((R, 1), 1e-10, dtype=raw.dtype)
    hi = torch.full((R, 1), 1e10, dtype=raw.dtype)
    tau = F.relu(torch.cat([lo, raw[..., 3] + noise, hi], dim=-1))
    e = torch.exp(-(0.5 * (tau[..., 1:] + tau[..., :-1])) * seg)
    T = torch.cumprod(torch.cat([torch.ones((R, 1), dtype=raw.dtype), e], dim=-1), dim=-1)
    w = (1 - e) * T[:, :-1]
    return w, tau, T


def weights_piecewise_constant(raw, z, rays_d, noise=0.0):
    """run_plnerf.py:504-513 (classic NeRF).  Returns weights [R,S]."""
    R = raw.shape[0]
    seg = z[..., 1:] - z[..., :-1]
    seg = torch.cat([seg, torch.full_like(seg[..., :1], 1e10)], dim=-1)
    seg = seg * torch.norm(rays_d[..., None, :], dim=-1)
    alpha = 1.0 - torch.exp(-F.relu(raw[..., 3] + noise) * seg)
    trans = torch.cumprod(
        torch.cat([torch.ones((R, 1), dtype=raw.dtype), 1.0 - alpha + 1e-10], dim=-1), dim=-1)[:, :-1]
    return alpha * trans


# ----------------------------------------------------------------------------
# a7: raw2outputs (run_plnerf.py:553-624)
# ----------------------------------------------------------------------------
def raw2outputs(raw, z, near, far, rays_d, mode, color_mode, raw_noise_std=0.0,
                pytest=False, white_bkgd=False, farcolorfix=False, noise=None):
    """Returns the reference 7-tuple (rgb_map, disp_map, acc_map, weights,
    depth_map, tau|None, T|None).  `noise` (optional tensor) overrides the random
    draw so HIP-vs-oracle tests can inject identical noise."""
    colour = torch.sigmoid(raw[..., :3])
    if noise is None:
        noise = 0.0
        if raw_noise_std > 0.0:
            noise = torch.randn(raw[..., 3].shape) * raw_noise_std          # line 570
            if pytest:                                                       # 573-576 (uniform!)
                np.random.seed(0)
                noise = torch.Tensor(np.random.rand(*list(raw[..., 3].shape)) * raw_noise_std)
    if mode == "linear":
        w, tau, T = weights_piecewise_linear(raw, z, near, far, rays_d, noise)
        first, last = colour[:, :1, :], colour[:, -1:, :]
        if color_mode == "midpoint":                                         # 581-591
            tail = torch.zeros_like(last) if farcolorfix else last
            padded = torch.cat([first, colour, tail], dim=1)
            rgb_map = torch.sum(w[..., None] * (0.5 * (padded[:, 1:] + padded[:, :-1])), dim=-2)
        elif color_mode == "left":                                           # 593-596
            padded = torch.cat([first, colour], dim=1)
            rgb_map = torch.sum(w[..., None] * padded, dim=-2)
        else:
            raise ValueError("color_mode must be 'midpoint' or 'left'")
        knots = torch.cat([near, z, far], dim=-1)
        depth_map = torch.sum(w * (0.5 * (knots[..., 1:] + knots[..., :-1])), dim=-1)   # 602-605
    elif mode == "constant":                                                 # 607-615
        w = weights_piecewise_constant(raw, z, rays_d, noise)
        rgb_map = torch.sum(w[..., None] * colour, dim=-2)
        depth_map = torch.sum(w * z, dim=-1)
        tau, T = None, None
    else:
        raise ValueError("mode must be 'linear' or 'constant'")
    acc_map = torch.sum(w, dim=-1)
    disp_map = 1.0 / torch.max(1e-10 * torch.ones_like(depth_map), depth_map / acc_map)   # 617
    if white_bkgd:
        rgb_map = rgb_map + (1.0 - acc_map[..., None])                       # 621-622
    return rgb_map, disp_map, acc_map, w, depth_map, tau, T


# ----------------------------------------------------------------------------
# a8: piecewise-constant inverse-CDF sampler (run_nerf_helpers.py:241-284)
# ----------------------------------------------------------------------------
def _draw_u(shape_prefix, n, det, pytest, u):
    if u is not None:
        return u.contiguous()
    if pytest:                                                               # 256-264
        np.random.seed(0)
        if det:
            u = np.broadcast_to(np.linspace(0.0, 1.0, n), list(shape_prefix) + [n])
        else:
            u = np.random.rand(*(list(shape_prefix) + [n]))
        return torch.Tensor(u).contiguous()
    if det:                                                                  # 249-251
        return torch.linspace(0.0, 1.0, steps=n).expand(list(shape_prefix) + [n]).contiguous()
    return torch.rand(list(shape_prefix) + [n])


def sample_pdf(bins, weights, n, det=False, pytest=False, u=None, return_inds=False):
    """bins [R,B], weights [R,B-1] -> samples [R,n] (+ searchsorted inds)."""
    w = weights + 1e-5
    pdf = w / torch.sum(w, dim=-1, keepdim=True)
    cdf = torch.cat([torch.zeros_like(pdf[..., :1]), torch.cumsum(pdf, dim=-1)], dim=-1)
    u = _draw_u(cdf.shape[:-1], n, det, pytest, u)
    inds = torch.searchsorted(cdf, u, right=True)
    below = torch.clamp(inds - 1, min=0)
    above = torch.clamp(inds, max=cdf.shape[-1] - 1)
    c0, c1 = torch.gather(cdf, -1, below), torch.gather(cdf, -1, above)
    b0, b1 = torch.gather(bins, -1, below), torch.gather(bins, -1, above)
    denom = c1 - c0
    denom = torch.where(denom < 1e-5, torch.ones_like(denom), denom)
    t = (u - c0) / denom
    samples = b0 + t * (b1 - b0)
    return (samples, inds) if return_inds else samples


# ----------------------------------------------------------------------------
# a9: exact inverse CDF under piecewise-linear density
#     (run_nerf_helpers.py:340-445)
# ----------------------------------------------------------------------------
def _invert_linear_segment(s0, s1, T0, tau0, tau1, u, eps, rising):
    """Solve T0 * exp(-(tau0 t + (tau1-tau0) t^2 / (2 (s1-s0)))) = 1-u for t, with
    the reference's epsilon guards (340-349 rising, 352-361 falling)."""
    e = torch.full_like(T0, eps)
    log_term = -torch.log(torch.max(e, (1 - u) / torch.max(e, T0)))
    span = torch.max(e, s1 - s0)
    if rising:
        disc = tau0 ** 2 + (2 * (tau1 - tau0) * log_term) / span
        t = ((s1 - s0) * (-tau0 + torch.sqrt(torch.max(e, disc)))) / torch.max(e, tau1 - tau0)
    else:
        disc = tau0 ** 2 - (2 * (tau0 - tau1) * log_term) / span
        t = ((s1 - s0) * (tau0 - torch.sqrt(torch.max(e, disc)))) / torch.max(e, tau0 - tau1)
    t = torch.clamp(t, e, s1 - s0)      # torch semantics: min applied first, then max
    return s0 + t


def sample_pdf_reformulation(z, weights, tau, T, near, far, n, det=False, pytest=False,
                             quad_solution_v2=False, zero_threshold=1e-4, epsilon_=1e-3,
                             u=None, return_inds=False):
    """z [R,S], weights [R,S+1], tau/T [R,S+2] -> (samples, T_below, tau_below,
    bin_below).  The cdf is the un-normalised cumsum of weights with its last
    entry forced to 1 (370-374).  H4: with u == 1.0 the reference
    indexes past tau_diff and raises; here `below` is clamped to S so the call is
    defined (the HIP kernel does the same)."""
    knots = torch.cat([near, z, far], dim=-1)
    cdf = torch.cat([torch.zeros_like(weights[..., :1]), torch.cumsum(weights, dim=-1)], dim=-1)
    cdf[:, -1] = 1.0
    u = _draw_u(cdf.shape[:-1], n, det, pytest, u)
    inds = torch.searchsorted(cdf, u, right=True)
    below = torch.clamp(inds - 1, min=0)
    above = torch.clamp(inds, max=cdf.shape[-1] - 1)
    s0, s1 = torch.gather(knots, -1, below), torch.gather(knots, -1, above)
    T0 = torch.gather(T, -1, below)
    tau0, tau1 = torch.gather(tau, -1, below), torch.gather(tau, -1, above)
    dtau = tau[..., 1:] - tau[..., :-1]
    d = torch.gather(dtau, -1, torch.clamp(below, max=dtau.shape[-1] - 1))   # 411-414 (+H4 clamp)
    zt = zero_threshold
    out = torch.where((d < zt) & (d > -zt), s0, torch.full_like(s0, -1.0))    # 425
    out = torch.where(d >= zt, _invert_linear_segment(s0, s1, T0, tau0, tau1, u, epsilon_, True), out)
    out = torch.where(d <= -zt, _invert_linear_segment(s0, s1, T0, tau0, tau1, u, epsilon_, False), out)
    out = torch.where(torch.isnan(out), s0, out)                              # 432
    if return_inds:
        return out, T0, tau0, s0, inds
    return out, T0, tau0, s0


# ----------------------------------------------------------------------------
# a1: render_rays (run_plnerf.py:627-758)
# ----------------------------------------------------------------------------
def stratified_z(near, far, n_samples, lindisp=False, perturb=0.0, pytest=False, t_rand=None):
    """run_plnerf.py:683-705."""
    R = near.shape[0]
    t = torch.linspace(0.0, 1.0, steps=n_samples)
    if not lindisp:
        z = near * (1.0 - t) + far * t
    else:
        z = 1.0 / (1.0 / near * (1.0 - t) + 1.0 / far * t)
    z = z.expand([R, n_samples])
    if perturb > 0.0:
        mid = 0.5 * (z[..., 1:] + z[..., :-1])
        hi = torch.cat([mid, z[..., -1:]], -1)
        lo = torch.cat([z[..., :1], mid], -1)
        if t_rand is None:
            t_rand = torch.rand(z.shape)
            if pytest:
                np.random.seed(0)
                t_rand = torch.Tensor(np.random.rand(*list(z.shape)))
        z = lo + (hi - lo) * t_rand
    return z


def render_rays(ray_batch, sd_coarse, sd_fine, N_samples, mode, color_mode, retraw=False,
                lindisp=False, perturb=0.0, N_importance=0, white_bkgd=False, raw_noise_std=0.0,
                pytest=False, zero_tol=1e-4, epsilon=1e-3, farcolorfix=False, constant_init=False,
                t_rand=None, u=None, return_internals=False):
    """ray_batch [R,11] = o(3) d(3) near far viewdir(3).  sd_* are state dicts.
    Optional t_rand/u inject the random draws (for HIP-vs-oracle comparisons on
    identical randomness)."""
    rays_o, rays_d = ray_batch[:, 0:3], ray_batch[:, 3:6]
    viewdirs = ray_batch[:, -3:]
    near, far = ray_batch[:, 6:7], ray_batch[:, 7:8]
    z = stratified_z(near, far, N_samples, lindisp, perturb, pytest, t_rand)
    pts = rays_o[..., None, :] + rays_d[..., None, :] * z[..., :, None]
    if constant_init:
        mode = "constant"                                                     # 710-711
    raw = query_network(sd_coarse, pts, viewdirs)
    rgb, disp, acc, w, depth, tau, T = raw2outputs(raw, z, near, far, rays_d, mode, color_mode,
                                                   raw_noise_std, pytest, white_bkgd, farcolorfix)
    internals = {"z_coarse": z, "raw_coarse": raw, "weights_coarse": w, "tau_coarse": tau, "T_coarse": T}
    ret = {}
    if N_importance > 0:
        coarse = (rgb, disp, acc, depth)
        if mode == "linear":
            z_new, _, _, _, inds = sample_pdf_reformulation(z, w, tau, T, near, far, N_importance,
                                                            det=(perturb == 0.0), pytest=pytest,
                                                            zero_threshold=zero_tol, epsilon_=epsilon, u=u,
                                                            return_inds=True)
            internals["inds"] = inds
        else:
            z_mid = 0.5 * (z[..., 1:] + z[..., :-1])
            z_new = sample_pdf(z_mid, w[..., 1:-1], N_importance, det=(perturb == 0.0),
                               pytest=pytest, u=u)
        z_new = z_new.detach()
        z_new = torch.clamp(z_new, near, far)                                 # 731
        z, _ = torch.sort(torch.cat([z, z_new], -1), -1)                      # 734
        pts = rays_o[..., None, :] + rays_d[..., None, :] * z[..., :, None]
        raw = query_network(sd_fine if sd_fine is not None else sd_coarse, pts, viewdirs)
        rgb, disp, acc, w, depth, tau, T = raw2outputs(raw, z, near, far, rays_d, mode, color_mode,
                                                       raw_noise_std, pytest, white_bkgd, farcolorfix)
        internals.update({"z_samples": z_new, "z_fine": z, "weights_fine": w})
    ret.update({"rgb_map": rgb, "disp_map": disp, "acc_map": acc, "depth_map": depth})
    if retraw:
        ret["raw"] = raw
    if N_importance > 0:
        ret["rgb0"], ret["disp0"], ret["acc0"], ret["depth0"] = coarse[0], coarse[1], coarse[2], coarse[3]
        ret["z_std"] = torch.std(z_new, dim=-1, unbiased=False)
    if return_internals:
        return ret, internals
    return ret


def fine_stage(ray_batch, sd_fine, z_fine, mode="linear", color_mode="midpoint", white_bkgd=False,
               raw_noise_std=0.0, pytest=False, farcolorfix=False, depth_variant=False):
    """The second half of render_rays on GIVEN merged depths z_fine [R,S] (run_plnerf.py:735-752;
    depth variant: run_nerf_sample_based_depth.py:907-921): positions, the fine network, raw2outputs.
    For per-stage comparisons on identical samples.  Returns a dict with raw, the
    maps, weights, tau, T."""
    rays_o, rays_d = ray_batch[:, 0:3], ray_batch[:, 3:6]
    viewdirs = ray_batch[:, 8:11]
    near, far = ray_batch[:, 6:7], ray_batch[:, 7:8]
    pts = rays_o[..., None, :] + rays_d[..., None, :] * z_fine[..., :, None]
    raw = query_network_depth(sd_fine, pts, viewdirs) if depth_variant else query_network(sd_fine, pts, viewdirs)
    rgb, disp, acc, w, depth, tau, T = raw2outputs(raw, z_fine, near, far, rays_d, mode, color_mode,
                                                   raw_noise_std, pytest, white_bkgd, farcolorfix)
    return {"raw": raw, "rgb_map": rgb, "disp_map": disp, "acc_map": acc, "depth_map": depth,
            "weights": w, "tau": tau, "T": T}


# ----------------------------------------------------------------------------
# a12: ray generation (run_nerf_helpers.py:162-201)
# ----------------------------------------------------------------------------
def get_rays(H, W, K, c2w):
    """Pinhole rays, no half-pixel offset; camera looks down -z (line 166)."""
    px = torch.linspace(0, W - 1, W)[None, :].expand(H, W)
    py = torch.linspace(0, H - 1, H)[:, None].expand(H, W)
    cam = torch.stack([(px - K[0][2]) / K[0][0], -(py - K[1][2]) / K[1][1], -torch.ones_like(px)], -1)
    d = torch.sum(cam[..., None, :] * c2w[:3, :3], -1)
    o = c2w[:3, -1].expand(d.shape)
    return o, d


def ndc_rays(H, W, focal, near, rays_o, rays_d):
    """run_nerf_helpers.py:184-201."""
    t = -(near + rays_o[..., 2]) / rays_d[..., 2]
    o = rays_o + t[..., None] * rays_d
    sx, sy = -1.0 / (W / (2.0 * focal)), -1.0 / (H / (2.0 * focal))
    o_ndc = torch.stack([sx * o[..., 0] / o[..., 2], sy * o[..., 1] / o[..., 2],
                         1.0 + 2.0 * near / o[..., 2]], -1)
    d_ndc = torch.stack([sx * (rays_d[..., 0] / rays_d[..., 2] - o[..., 0] / o[..., 2]),
                         sy * (rays_d[..., 1] / rays_d[..., 2] - o[..., 1] / o[..., 2]),
                         -2.0 * near / o[..., 2]], -1)
    return o_ndc, d_ndc


def pose_spherical(theta_deg, phi_deg, radius):
    """Camera-to-world of the Blender orbit (load_blender.py:13-34): translate
    along z, rotate about x by phi, about y by theta, then swap axes."""
    def trans(t):
        m = np.eye(4); m[2, 3] = t; return m

    def rot_phi(p):
        c, s = np.cos(p), np.sin(p)
        return np.array([[1, 0, 0, 0], [0, c, -s, 0], [0, s, c, 0], [0, 0, 0, 1.0]])

    def rot_theta(th):
        c, s = np.cos(th), np.sin(th)
        return np.array([[c, 0, -s, 0], [0, 1, 0, 0], [s, 0, c, 0], [0, 0, 0, 1.0]])
    m = rot_theta(theta_deg / 180.0 * np.pi) @ rot_phi(phi_deg / 180.0 * np.pi) @ trans(radius)
    m = np.array([[-1, 0, 0, 0], [0, 0, 1, 0], [0, 1, 0, 0], [0, 0, 0, 1.0]]) @ m
    return torch.from_numpy(m.astype(np.float32))


def pack_ray_batch(rays_o, rays_d, near, far):
    """render()'s packing (run_plnerf.py:143-164): [o, d, near, far, d/|d|]."""
    o = rays_o.reshape(-1, 3).float()
    d = rays_d.reshape(-1, 3).float()
    v = d / torch.norm(d, dim=-1, keepdim=True)
    n = near * torch.ones_like(d[..., :1])
    f = far * torch.ones_like(d[..., :1])
    return torch.cat([o, d, n, f, v], -1)


# ----------------------------------------------------------------------------
# a13: one optimisation step (run_plnerf.py:1283-1316)
# ----------------------------------------------------------------------------
def train_step(sd_coarse, sd_fine, ray_batch, target, render_kwargs, lr=5e-4, adam_state=None, return_psnr=False):
    """loss = mse(rgb_map, target) + mse(rgb0, target); Adam(0.9, 0.999) on both
    nets.  Parameters are updated IN PLACE; returns (loss, grads_coarse,
    grads_fine[, psnr]) -- psnr = mse2psnr(img2mse(rgb_map, target)) as the reference
    prints it (run_plnerf.py:1288-1290) when return_psnr.  `adam_state` persists
    optimiser state across calls."""
    params_c = [p.requires_grad_(True) for p in sd_coarse.values()]
    params_f = [p.requires_grad_(True) for p in sd_fine.values()]
    if adam_state is None or "opt_f" not in adam_state:
        opt_f = torch.optim.Adam(params_f, lr=lr, betas=(0.9, 0.999))
        opt_c = torch.optim.Adam(params_c, lr=lr, betas=(0.9, 0.999))
        if adam_state is not None:
            adam_state["opt_f"], adam_state["opt_c"] = opt_f, opt_c
    else:
        opt_f, opt_c = adam_state["opt_f"], adam_state["opt_c"]
        for opt in (opt_f, opt_c):     # the caller's schedule (run_plnerf.py:1307-1315)
            for group in opt.param_groups:
                group["lr"] = lr
    ret = render_rays(ray_batch, sd_coarse, sd_fine, retraw=True, **render_kwargs)
    opt_f.zero_grad()
    opt_c.zero_grad()
    img_loss = torch.mean((ret["rgb_map"] - target) ** 2)
    loss = img_loss
    if "rgb0" in ret:
        loss = loss + torch.mean((ret["rgb0"] - target) ** 2)
    loss.backward()
    g_c = {k: (v.grad.clone() if v.grad is not None else torch.zeros_like(v)) for k, v in sd_coarse.items()}
    g_f = {k: (v.grad.clone() if v.grad is not None else torch.zeros_like(v)) for k, v in sd_fine.items()}
    opt_f.step()
    opt_c.step()
    if return_psnr:
        return loss.detach(), g_c, g_f, -10.0 * torch.log10(img_loss.detach())
    return loss.detach(), g_c, g_f


def synthetic_blender_rays(n_rays, seed=0, near=2.0, far=6.0, H=800, W=800, theta=30.0):
    """Synthetic 800x800 Blender-style ray batch: focal
    1111.111, pose_spherical(theta,-30,4), pixels drawn without replacement."""
    focal = 0.5 * W / math.tan(0.5 * 0.6911112070083618)
    K = [[focal, 0, 0.5 * W], [0, focal, 0.5 * H], [0, 0, 1]]
    c2w = pose_spherical(theta, -30.0, 4.0)[:3, :4]
    o, d = get_rays(H, W, K, c2w)
    rng = np.random.default_rng(seed)
    pix = torch.from_numpy(rng.choice(H * W, n_rays, replace=False))
    o, d = o.reshape(-1, 3)[pix], d.reshape(-1, 3)[pix]
    target = torch.from_numpy(rng.random((n_rays, 3), dtype=np.float32))
    return pack_ray_batch(o, d, near, far), target


# ============================================================================
# Depth-supervised variant of the path.  Restates
# depth_supervised_exps/run_nerf_sample_based_depth.py and
# depth_supervised_exps/model/run_nerf_helpers.py; pinned by fixture G8.
# ============================================================================
DEPTH_XYZ_FREQS = 9     # --multires default (run_nerf_sample_based_depth.py:1306)
DEPTH_DIR_FREQS = 0     # --multires_views default (:1308): identity only -> 3 channels


def param_shapes_depth(xyz_freqs=DEPTH_XYZ_FREQS, dir_freqs=DEPTH_DIR_FREQS):
    """Same module tree as param_shapes() with input_ch = 3 + 6*9 = 57 and input_ch_views = 3
    (model/run_nerf_helpers.py:164-179 with the depth script's default flags)."""
    xyz, dirc = 3 + 6 * xyz_freqs, 3 + 6 * dir_freqs
    out = []
    for name, shape in param_shapes():
        if name == "pts_linears.0.weight":
            shape = (WIDTH, xyz)
        elif name == f"pts_linears.{SKIP_AFTER + 1}.weight":
            shape = (WIDTH, WIDTH + xyz)
        elif name == "views_linears.0.weight":
            shape = (WIDTH // 2, WIDTH + dirc)
        out.append((name, shape))
    return out


def closed_form_state_dict_depth(seed=0, sharpen=False):
    """closed_form_state_dict's sine hash on the depth variant's shapes."""
    shapes = param_shapes_depth()
    table = dict(shapes)
    sd = {}
    for li, (name, shape) in enumerate(shapes):
        fan_in = shape[1] if len(shape) == 2 else table[name.replace("bias", "weight")][1]
        bound = 1.0 / math.sqrt(fan_in)
        idx = np.arange(int(np.prod(shape)), dtype=np.float64)
        h = np.sin(12.9898 * (idx + 1.0) + 78.233 * (li + 1.0) + 37.719 * (seed + 1.0)) * 43758.5453
        frac = h - np.floor(h)
        sd[name] = torch.from_numpy(((2.0 * frac - 1.0) * bound).astype(np.float32).reshape(shape))
    if sharpen:
        sd["alpha_linear.weight"] = sd["alpha_linear.weight"] * 30.0
        sd["alpha_linear.bias"] = sd["alpha_linear.bias"] + 0.5
    return sd


def positional_encoding_pi(x, n_freqs):
    """model/run_nerf_helpers.py:100-130: p_fn(x * np.pi * freq), i.e. (x * pi) * 2^k in x's dtype."""
    blocks = [x]
    for k in range(n_freqs):
        xs = x * np.pi * float(2 ** k)
        blocks.append(torch.sin(xs))
        blocks.append(torch.cos(xs))
    return torch.cat(blocks, dim=-1)


def nerf_mlp_depth(sd, embedded):
    """model/run_nerf_helpers.py:181-205: the same trunk, input widths read off the weights, and
    softplus(beta=10) on the density channel (:200)."""
    xyz_ch = sd["pts_linears.0.weight"].shape[1]
    enc_xyz, enc_dir = embedded[..., :xyz_ch], embedded[..., xyz_ch:]
    h = enc_xyz
    for i in range(DEPTH):
        h = F.relu(F.linear(h, sd[f"pts_linears.{i}.weight"], sd[f"pts_linears.{i}.bias"]))
        if i == SKIP_AFTER:
            h = torch.cat([enc_xyz, h], dim=-1)
    alpha = F.linear(h, sd["alpha_linear.weight"], sd["alpha_linear.bias"])
    feat = F.linear(h, sd["feature_linear.weight"], sd["feature_linear.bias"])
    hv = F.relu(F.linear(torch.cat([feat, enc_dir], dim=-1), sd["views_linears.0.weight"],
                         sd["views_linears.0.bias"]))
    rgb = F.linear(hv, sd["rgb_linear.weight"], sd["rgb_linear.bias"])
    return torch.cat([rgb, F.softplus(alpha, beta=10)], dim=-1)


def query_network_depth(sd, pts, viewdirs, bb_center=0.0, bb_scale=1.0, xyz_freqs=DEPTH_XYZ_FREQS,
                        dir_freqs=DEPTH_DIR_FREQS, netchunk=65536, embedded_cam=None):
    """run_nerf_sample_based_depth.py:52-68.  embedded_cam (a vector, :1122-1123; empty by default): repeated on every
    row behind the direction encoding (:64); the view layer's weight then has that many more input columns
    (model/run_nerf_helpers.py:164-170).  Pinned to the reference by fixture G8c."""
    R, S = pts.shape[0], pts.shape[1]
    flat = (pts.reshape(-1, 3) - bb_center) * bb_scale
    emb = positional_encoding_pi(flat, xyz_freqs)
    dirs = viewdirs[:, None, :].expand(R, S, 3).reshape(-1, 3)
    emb = torch.cat([emb, positional_encoding_pi(dirs, dir_freqs)], dim=-1)
    if embedded_cam is not None and embedded_cam.numel() > 0:
        emb = torch.cat([emb, embedded_cam.reshape(1, -1).expand(emb.shape[0], embedded_cam.numel())], dim=-1)
    outs = [nerf_mlp_depth(sd, emb[i:i + netchunk]) for i in range(0, emb.shape[0], netchunk)]
    return torch.cat(outs, 0).reshape(R, S, 4)


def _draw_u_depth(R, n, det, pytest, load_u):
    """model/run_nerf_helpers.py:619-638."""
    if load_u is not None:
        return load_u
    if det:
        u = torch.linspace(0.0, 1.0, steps=n).expand(R, n)
    else:
        u = torch.rand(R, n)
    if pytest:
        np.random.seed(0)
        if det:
            u = torch.Tensor(np.broadcast_to(np.linspace(0.0, 1.0, n), [R, n]).copy())
        else:
            u = torch.Tensor(np.random.rand(R, n))
    return u


def render_rays_depth(ray_batch, sd_coarse, sd_fine, N_samples, mode, color_mode, perturb=0.0,
                      N_importance=0, white_bkgd=False, raw_noise_std=0.0, pytest=False, cached_u=None,
                      bb_center=0.0, bb_scale=1.0, t_rand=None, u_fine=None, return_internals=False):
    """run_nerf_sample_based_depth.py:792-958, is_joint False.  `pred_hyp` stays attached: to the final
    weights' tau and T in mode 'linear' (:923-934), to the final weights in mode 'constant'.  t_rand / u_fine
    inject the stratified draw and the importance draw for HIP-vs-oracle comparisons."""
    rays_o, rays_d = ray_batch[:, 0:3], ray_batch[:, 3:6]
    viewdirs = ray_batch[:, 8:11]
    near, far = ray_batch[:, 6:7], ray_batch[:, 7:8]
    z = stratified_z(near, far, N_samples, False, perturb, pytest, t_rand)
    pts = rays_o[..., None, :] + rays_d[..., None, :] * z[..., :, None]
    raw = query_network_depth(sd_coarse, pts, viewdirs, bb_center, bb_scale)
    rgb, disp, acc, w, depth, tau, T = raw2outputs(raw, z, near, far, rays_d, mode, color_mode, raw_noise_std,
                                                   pytest, white_bkgd)
    R = ray_batch.shape[0]
    ret, internals = {}, {}

    def draw(z, w, tau, T, n, u=None, det=False, pyt=False):
        if mode == "linear":
            return sample_pdf_reformulation(z, w, tau, T, near, far, n, det=det, pytest=pyt, u=u)[0]
        return sample_pdf(0.5 * (z[..., 1:] + z[..., :-1]), w[..., 1:-1], n, det=det, pytest=pyt, u=u)
    if N_importance == 0:
        u = _draw_u_depth(R, N_samples, perturb == 0.0, pytest, None)
        hyp = draw(z, w, tau, T, N_samples, u=u)
    else:
        coarse = (rgb, disp, acc, depth, z, w)
        internals = {"tau_coarse": tau, "T_coarse": T, "weights_coarse": w, "raw_coarse": raw}      # (for per-stage comparisons)
        z_new = draw(z, w, tau, T, N_importance, u=u_fine, det=(perturb == 0.0), pyt=pytest).detach()
        z_new = torch.clamp(z_new, near, far)
        z, _ = torch.sort(torch.cat([z, z_new], -1), -1)
        pts = rays_o[..., None, :] + rays_d[..., None, :] * z[..., :, None]
        raw = query_network_depth(sd_fine if sd_fine is not None else sd_coarse, pts, viewdirs, bb_center, bb_scale)
        rgb, disp, acc, w, depth, tau, T = raw2outputs(raw, z, near, far, rays_d, mode, color_mode, raw_noise_std,
                                                       pytest, white_bkgd)
        u = _draw_u_depth(R, N_importance, perturb == 0.0, pytest, cached_u)
        hyp = draw(z, w, tau, T, N_importance, u=u)
        ret.update({"rgb0": coarse[0], "disp0": coarse[1], "acc0": coarse[2], "depth0": coarse[3],
                    "z_vals0": coarse[4], "weights0": coarse[5],
                    "z_std": torch.std(hyp, dim=-1, unbiased=False)})
    ret.update({"rgb_map": rgb, "disp_map": disp, "acc_map": acc, "depth_map": depth, "z_vals": z,
                "weights": w[..., 1:] if mode == "linear" else w, "pred_hyp": hyp, "u": u, "raw": raw})
    if return_internals:
        internals.update({"z_samples": z_new, "tau": tau, "T": T, "weights_full": w} if N_importance > 0 else {})
        return ret, internals
    return ret


def compute_space_carving_loss(pred_depth, target_hypothesis, is_joint=False, mask=None, norm_p=2, threshold=0.0):
    """model/run_nerf_helpers.py:52-86."""
    n_points = pred_depth.shape[1]
    target = target_hypothesis.repeat(1, 1, n_points) if target_hypothesis.shape[-1] == 1 else target_hypothesis
    dist = torch.norm(pred_depth.unsqueeze(-1) - target.unsqueeze(-1), p=norm_p, dim=-1)
    if mask is not None:
        dist = dist * mask.unsqueeze(0).repeat(dist.shape[0], 1).unsqueeze(-1)
    if threshold > 0:
        dist = torch.where(dist < threshold, torch.zeros(()), dist)
    if is_joint:
        return torch.mean(torch.min(torch.mean(dist, axis=1), axis=0)[0], axis=-1)
    return torch.mean(torch.mean(torch.min(dist, dim=0)[0], dim=-1))


def depth_train_step(sd_coarse, sd_fine, ray_batch, target_s, target_h, render_kwargs, space_carving_weight=0.007,
                     lr=5e-4, mask=None, adam_state=None):
    """run_nerf_sample_based_depth.py:1126-1157: loss = mse(rgb) + w * space_carving(pred_hyp, target_h) +
    mse(rgb0); clip_grad_value_(0.1); ONE Adam over both networks.  Parameters are updated in place; returns
    (loss, space_carving_loss, grads_coarse, grads_fine) with the gradients BEFORE clipping."""
    params = [p.requires_grad_(True) for p in list(sd_coarse.values()) + list(sd_fine.values())]
    if adam_state is None or "opt" not in adam_state:
        opt = torch.optim.Adam(params, lr=lr, betas=(0.9, 0.999))
        if adam_state is not None:
            adam_state["opt"] = opt
    else:
        opt = adam_state["opt"]
    ret = render_rays_depth(ray_batch, sd_coarse, sd_fine, **render_kwargs)
    opt.zero_grad()
    loss = torch.mean((ret["rgb_map"] - target_s) ** 2)
    sc = compute_space_carving_loss(ret["pred_hyp"], target_h, mask=mask)
    loss = loss + space_carving_weight * sc
    if "rgb0" in ret:
        loss = loss + torch.mean((ret["rgb0"] - target_s) ** 2)
    loss.backward()
    g_c = {k: (v.grad.clone() if v.grad is not None else torch.zeros_like(v)) for k, v in sd_coarse.items()}
    g_f = {k: (v.grad.clone() if v.grad is not None else torch.zeros_like(v)) for k, v in sd_fine.items()}
    torch.nn.utils.clip_grad_value_(params, 0.1)
    opt.step()
    return loss.detach(), sc.detach(), g_c, g_f
