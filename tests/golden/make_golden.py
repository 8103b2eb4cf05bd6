"""Generate the golden fixtures tests/golden/*.npz from the REFERENCE itself.

Run in the build container only (it imports /root/reference read-only through
_ref_import.py):

    python tests/golden/make_golden.py

Fixtures are data only: inputs and the reference's outputs.  Network weights
come from oracle.plnerf_oracle.closed_form_state_dict (an RNG-free recipe), so
they are not stored.  Fixture ids follow SURVEY.md section 8c (G1..G8); G8b = G8's step at configs[4]'s 128+64 sampling, G8c = the
camera code (input_ch_cam = 4) through the reference's network and run_network, G9 = the
reference's checkpoint file and what the reference computes after re-loading it.
"""
import os
import sys
import tempfile
from argparse import Namespace

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)

from _ref_import import import_reference  # noqa: E402
from oracle import plnerf_oracle as orc  # noqa: E402

R_, H_ = import_reference()
torch.set_num_threads(8)


def npz(name, **arrays):
    out = {}
    for k, v in arrays.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        out[k] = np.asarray(v)
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print(f"wrote {path} ({os.path.getsize(path)/1024:.1f} KiB)")


def make_args(N_samples, N_importance, mode, white_bkgd=True, dataset="blender", raw_noise_std=0.0):
    d = tempfile.mkdtemp()
    os.makedirs(os.path.join(d, "exp"))
    return Namespace(multires=10, i_embed=0, use_viewdirs=True, multires_views=4,
                     N_importance=N_importance, N_samples=N_samples, netdepth=8, netwidth=256,
                     netdepth_fine=8, netwidth_fine=256, netchunk=65536, lrate=5e-4,
                     coarse_lrate=5e-4, ft_path=None, ckpt_dir=d, expname="exp", no_reload=True,
                     perturb=1.0, white_bkgd=white_bkgd, raw_noise_std=raw_noise_std, mode=mode,
                     color_mode="midpoint", dataset=dataset, no_ndc=False, lindisp=False)


def ref_nets(args, seed_c=0, seed_f=1, sharpen=True):
    kw_train, kw_test, start, grad_vars, opt, opt_c = R_.create_nerf(args)
    kw_train["network_fn"].load_state_dict(orc.closed_form_state_dict(seed_c, sharpen))
    kw_train["network_fine"].load_state_dict(orc.closed_form_state_dict(seed_f, sharpen))
    return kw_train, kw_test, opt, opt_c


def scene_rays(n, seed, near=2.0, far=6.0):
    batch, target = orc.synthetic_blender_rays(n, seed=seed, near=near, far=far)
    return batch, target


# ---------------------------------------------------------------- G1: PE + MLP
def g1():
    g = torch.Generator().manual_seed(11)
    R, S = 6, 40
    pts = (torch.rand(R, S, 3, generator=g) * 2 - 1) * 3.0
    vd = torch.randn(R, 3, generator=g)
    vd = vd / vd.norm(dim=-1, keepdim=True)
    args = make_args(64, 128, "linear")
    out = {}
    for tag, sharpen in (("plain", False), ("sharp", True)):
        kw, _, _, _ = ref_nets(args, 0, 1, sharpen)
        with torch.no_grad():
            raw = kw["network_query_fn"](pts, vd, kw["network_fn"])
            emb = torch.cat([H_.get_embedder(10, 0)[0](pts.reshape(-1, 3)),
                             H_.get_embedder(4, 0)[0](vd[:, None].expand(R, S, 3).reshape(-1, 3))], -1)
            raw_emb = kw["network_fn"](emb)
        out[f"raw_{tag}"] = raw
        out[f"raw_from_embedded_{tag}"] = raw_emb
        out["embedded"] = emb
    npz("g1_mlp", pts=pts, viewdirs=vd, **out)


# ---------------------------------------------------------------- G2: raw2outputs
def quad_inputs(R, S, seed, near=2.0, far=6.0, end_at_far=False):
    g = torch.Generator().manual_seed(seed)
    raw = torch.randn(R, S, 4, generator=g)
    raw[..., 3] = raw[..., 3] * 3.0 + 0.5          # densities on both sides of zero
    raw[R // 2:, S // 3: S // 2, 3] += 25.0        # an opaque slab on half the rays
    z, _ = torch.sort(near + (far - near) * torch.rand(R, S, generator=g), -1)
    if end_at_far:
        z[:, -1] = far
    nr = torch.full((R, 1), near)
    fr = torch.full((R, 1), far)
    d = torch.randn(R, 3, generator=g) * 1.3
    return raw, z, nr, fr, d


def g2():
    out = {}
    case = 0
    for S in (64, 128, 192):
        for mode, cmode in (("linear", "midpoint"), ("linear", "left"), ("constant", "midpoint")):
            for wb in (False, True):
                raw, z, nr, fr, d = quad_inputs(12, S, 100 + case, end_at_far=(case % 4 == 3))
                res = R_.raw2outputs(raw, z, nr, fr, d, mode, cmode, 0.0, pytest=False, white_bkgd=wb)
                p = f"c{case}_"
                out.update({p + "raw": raw, p + "z": z, p + "near": nr, p + "far": fr, p + "rays_d": d,
                            p + "S": S, p + "mode": mode, p + "color_mode": cmode, p + "white_bkgd": wb,
                            p + "noise_std": 0.0})
                names = ["rgb_map", "disp_map", "acc_map", "weights", "depth_map", "tau", "T"]
                for nme, v in zip(names, res):
                    if v is not None:
                        out[p + nme] = v
                case += 1
    # noise (pytest=True -> uniform np.random.rand, seed 0) and farcolorfix
    for mode, ffix in (("linear", False), ("constant", False), ("linear", True)):
        raw, z, nr, fr, d = quad_inputs(12, 64, 100 + case)
        res = R_.raw2outputs(raw, z, nr, fr, d, mode, "midpoint", 1.0, pytest=True, white_bkgd=True,
                             farcolorfix=ffix)
        p = f"c{case}_"
        out.update({p + "raw": raw, p + "z": z, p + "near": nr, p + "far": fr, p + "rays_d": d,
                    p + "S": 64, p + "mode": mode, p + "color_mode": "midpoint", p + "white_bkgd": True,
                    p + "noise_std": 1.0, p + "farcolorfix": ffix})
        for nme, v in zip(["rgb_map", "disp_map", "acc_map", "weights", "depth_map", "tau", "T"], res):
            if v is not None:
                out[p + nme] = v
        case += 1
    out["n_cases"] = case
    npz("g2_raw2outputs", **out)


# ---------------------------------------------------------------- G3: sample_pdf (det)
class CaptureSearchsorted:
    def __enter__(self):
        self.orig = torch.searchsorted
        self.inds = []

        def wrapped(*a, **k):
            r = self.orig(*a, **k)
            self.inds.append(r.clone())
            return r
        torch.searchsorted = wrapped
        return self

    def __exit__(self, *exc):
        torch.searchsorted = self.orig


def g3():
    out = {}
    for ci, (B, N) in enumerate(((63, 128), (127, 64), (63, 64))):
        g = torch.Generator().manual_seed(300 + ci)
        R = 48
        bins, _ = torch.sort(2.0 + 4.0 * torch.rand(R, B, generator=g), -1)
        w = torch.rand(R, B - 1, generator=g) ** 4
        w[R // 3: 2 * R // 3] *= 1e-4                   # near-empty rays: pdf ~ uniform from the +1e-5
        w[: R // 6, 5:20] = 0.0                         # zero runs -> denom < 1e-5 branch
        w[-4:] = 0.0                                    # all-zero rays
        for det in (True, False):
            with CaptureSearchsorted() as cap:
                # det: the real torch.linspace draw (pytest=False); random: numpy seed-0 draw
                s = H_.sample_pdf(bins, w, N, det=det, pytest=not det)
            p = f"c{ci}_{'det' if det else 'rnd'}_"
            out.update({p + "samples": s, p + "inds": cap.inds[0]})
        out.update({f"c{ci}_bins": bins, f"c{ci}_weights": w, f"c{ci}_N": N})
    out["n_cases"] = 3
    npz("g3_sample_pdf", **out)


# ---------------------------------------------------------------- G4: PL sampler
def g4():
    out = {}
    for ci, (S, N) in enumerate(((64, 128), (128, 64))):
        raw, z, nr, fr, d = quad_inputs(40, S, 400 + ci)
        # engineered branches: equal neighbouring densities (|dtau| < zero_tol), a NaN density,
        # and near-coincident knots (s_r - s_l < epsilon)
        raw[0:6, 10:20, 3] = 0.75
        raw[6:10, :, 3] = -1.0                          # all tau == 0 -> constant branch everywhere
        z[10:14, 20:24] = z[10:14, 20:21] + torch.arange(4) * 1e-5
        z, _ = torch.sort(z, -1)
        _, _, _, w, _, tau, T = R_.raw2outputs(raw, z, nr, fr, d, "linear", "midpoint", 0.0,
                                               pytest=False, white_bkgd=True)
        with CaptureSearchsorted() as cap:
            s, Tb, taub, binb = H_.sample_pdf_reformulation(z, w, tau, T, nr, fr, N, det=False,
                                                            pytest=True, zero_threshold=1e-4,
                                                            epsilon_=1e-3)
        np.random.seed(0)
        u = torch.Tensor(np.random.rand(40, N))
        p = f"c{ci}_"
        out.update({p + "z": z, p + "weights": w, p + "tau": tau, p + "T": T, p + "near": nr, p + "far": fr,
                    p + "u": u, p + "N": N, p + "samples": s, p + "T_below": Tb, p + "tau_below": taub,
                    p + "bin_below": binb, p + "inds": cap.inds[0]})
    out["n_cases"] = 2
    npz("g4_sample_pl", **out)


# ---------------------------------------------------------------- G5: render_rays end to end
def g5():
    out = {}
    ci = 0
    for (Ns, Ni, mode, ndc) in ((64, 128, "linear", False), (128, 64, "linear", False),
                                (64, 128, "constant", False), (128, 64, "linear", True)):
        args = make_args(Ns, Ni, mode, white_bkgd=not ndc, dataset="llff" if ndc else "blender",
                         raw_noise_std=1.0 if ndc else 0.0)
        kw, _, _, _ = ref_nets(args, 0, 1, sharpen=True)
        R = 12
        if not ndc:
            batch, _ = scene_rays(R, seed=50 + ci)
            rays = (batch[:, 0:3], batch[:, 3:6])
            near, far, H, W, K = 2.0, 6.0, 800, 800, None
            res = R_.render(800, 800, [[1111.111, 0, 400], [0, 1111.111, 400], [0, 0, 1]], chunk=32768,
                            rays=rays, near=near, far=far, retraw=True, pytest=True,
                            **kw)
        else:
            g = torch.Generator().manual_seed(77)
            H, W, f = 378, 504, 407.0
            Kc = [[f, 0, W / 2], [0, f, H / 2], [0, 0, 1]]
            c2w = torch.eye(4)[:3, :4].clone()
            c2w[:, 3] = torch.tensor([0.05, -0.02, 0.1])
            o, dd = H_.get_rays(H, W, Kc, c2w)
            pix = torch.randperm(H * W, generator=g)[:R]
            rays = (o.reshape(-1, 3)[pix], dd.reshape(-1, 3)[pix])
            res = R_.render(H, W, Kc, chunk=32768, rays=rays, near=0.0, far=1.0, retraw=True, pytest=True,
                            **kw)
            near, far = 0.0, 1.0
        rgb, disp, acc, extras = res
        p = f"c{ci}_"
        out.update({p + "rays_o": rays[0], p + "rays_d": rays[1], p + "near": near, p + "far": far,
                    p + "N_samples": Ns, p + "N_importance": Ni, p + "mode": mode, p + "ndc": ndc,
                    p + "H": H, p + "W": W, p + "focal": 1111.111 if not ndc else 407.0,
                    p + "white_bkgd": not ndc, p + "raw_noise_std": 1.0 if ndc else 0.0,
                    p + "rgb_map": rgb, p + "disp_map": disp, p + "acc_map": acc})
        for k, v in extras.items():
            out[p + k] = v
        ci += 1
    out["n_cases"] = ci
    npz("g5_render_rays", **out)


# ---------------------------------------------------------------- G6: one training step
def sample_elems(t, stride=97):
    return t.detach().reshape(-1)[::stride].clone()


def g6():
    out = {}
    for ci, (Ns, Ni) in enumerate(((64, 128), (128, 64))):
        args = make_args(Ns, Ni, "linear")
        kw, _, opt, opt_c = ref_nets(args, 0, 1, sharpen=False)
        R = 24
        batch, target = scene_rays(R, seed=60 + ci)
        rays = (batch[:, 0:3], batch[:, 3:6])
        rgb, disp, acc, extras = R_.render(800, 800, [[1111.111, 0, 400], [0, 1111.111, 400], [0, 0, 1]],
                                           chunk=32768, rays=rays, near=2.0, far=6.0, retraw=True,
                                           pytest=True, **kw)
        opt.zero_grad()
        opt_c.zero_grad()
        loss = H_.img2mse(rgb, target) + H_.img2mse(extras["rgb0"], target)
        loss.backward()
        p = f"c{ci}_"
        out.update({p + "ray_batch": batch, p + "target": target, p + "N_samples": Ns, p + "N_importance": Ni,
                    p + "loss": loss.detach()})
        for net, tag in ((kw["network_fn"], "coarse"), (kw["network_fine"], "fine")):
            for name, prm in net.named_parameters():
                out[p + f"grad_{tag}_{name}_norm"] = prm.grad.norm()
                out[p + f"grad_{tag}_{name}_sample"] = sample_elems(prm.grad)
        opt.step()
        opt_c.step()
        for net, tag in ((kw["network_fn"], "coarse"), (kw["network_fine"], "fine")):
            for name, prm in net.named_parameters():
                out[p + f"param_{tag}_{name}_sample"] = sample_elems(prm)
    out["n_cases"] = 2
    out["sample_stride"] = 97
    npz("g6_train_step", **out)


# ---------------------------------------------------------------- G7: rays
def g7():
    H, W, f = 6, 8, 7.5
    K = [[f, 0, W / 2], [0, f, H / 2], [0, 0, 1]]
    c2w = orc.pose_spherical(37.0, -30.0, 4.0)[:3, :4]
    o, d = H_.get_rays(H, W, K, c2w)
    o2, d2 = H_.ndc_rays(H, W, f, 1.0, o, d)
    npz("g7_rays", H=H, W=W, focal=f, c2w=c2w, rays_o=o, rays_d=d, ndc_o=o2, ndc_d=d2)


# ---------------------------------------------------------------- G8: depth-supervised variant
def import_depth_reference():
    """depth_supervised_exps/run_nerf_sample_based_depth.py and its model package, imported read-only."""
    ddir = os.path.join("/root/reference", "depth_supervised_exps")
    if ddir not in sys.path:
        sys.path.insert(0, ddir)
    import run_nerf_sample_based_depth as D      # noqa: E402
    import model.run_nerf_helpers as DH          # noqa: E402
    return D, DH


def _depth_nets(DH, input_ch, input_ch_views):
    def net(seed):
        m = DH.NeRF(D=8, W=256, input_ch=input_ch, output_ch=5, skips=[4], input_ch_views=input_ch_views,
                    input_ch_cam=0, use_viewdirs=True)
        m.load_state_dict(orc.closed_form_state_dict_depth(seed, sharpen=True))
        return m
    return net(0), net(1)


def _depth_render_and_step(D, DH, coarse, fine, query, R, Ns, Ni, seed, n_hyp=3, sc_w=0.007):
    """render_rays of the depth script (with the attached pred_hyp) and one clipped Adam step over both networks."""
    out = {}
    batch, target = scene_rays(R, seed=seed)
    rng = np.random.default_rng(seed)
    target_h = torch.from_numpy(rng.uniform(2.0, 6.0, size=(n_hyp, R, 1)).astype(np.float32))
    params = list(coarse.parameters()) + list(fine.parameters())
    opt = torch.optim.Adam(params=params, lr=5e-4, betas=(0.9, 0.999))
    ret = D.render_rays(batch, True, coarse, query, Ns, "linear", "midpoint", embedded_cam=torch.tensor(()),
                        retraw=True, perturb=1.0, N_importance=Ni, network_fine=fine, raw_noise_std=0.0,
                        pytest=True, white_bkgd=True, is_joint=False, cached_u=None)
    for k in ("rgb_map", "disp_map", "acc_map", "depth_map", "z_vals", "weights", "pred_hyp", "u", "raw", "rgb0",
              "disp0", "acc0", "depth0", "z_vals0", "weights0", "z_std"):
        out["render_" + k] = ret[k]
    opt.zero_grad()
    img_loss = DH.img2mse(ret["rgb_map"], target)
    sc = DH.compute_space_carving_loss(ret["pred_hyp"], target_h, is_joint=False, norm_p=2, threshold=0.0, mask=None)
    loss = img_loss + sc_w * sc + DH.img2mse(ret["rgb0"], target)
    loss.backward()
    for m, tag in ((coarse, "coarse"), (fine, "fine")):
        for name, prm in m.named_parameters():
            out[f"grad_{tag}_{name}_norm"] = prm.grad.norm()
            out[f"grad_{tag}_{name}_sample"] = sample_elems(prm.grad)
    torch.nn.utils.clip_grad_value_(params, 0.1)
    opt.step()
    for m, tag in ((coarse, "coarse"), (fine, "fine")):
        for name, prm in m.named_parameters():
            out[f"param_{tag}_{name}_sample"] = sample_elems(prm)
    out.update({"ray_batch": batch, "target": target, "target_h": target_h, "N_samples": Ns, "N_importance": Ni,
                "space_carving_weight": sc_w, "loss": loss.detach(), "space_carving_loss": sc.detach(),
                "sample_stride": 97})
    return out


def g8b():
    """G8 at BASELINE.json configs[4]'s sampling: N_samples = 128, N_importance = 64 (depth-supervised, mode=linear)."""
    D, DH = import_depth_reference()
    embed_fn, input_ch = DH.get_embedder(9, 0)
    embeddirs_fn, input_ch_views = DH.get_embedder(0, 0)
    coarse, fine = _depth_nets(DH, input_ch, input_ch_views)

    def query(inputs, viewdirs, embedded_cam, fn):
        return D.run_network(inputs, viewdirs, embedded_cam, fn, embed_fn=embed_fn, embeddirs_fn=embeddirs_fn,
                             bb_center=0.0, bb_scale=1.0, netchunk=65536)
    npz("g8b_depth_variant_128_64", **_depth_render_and_step(D, DH, coarse, fine, query, R=16, Ns=128, Ni=64, seed=88))


def g8c():
    """The camera code of the depth-supervised script (input_ch_cam = 4: run_nerf_sample_based_depth.py:52-68, 1091-1093,
    1122-1123; model/run_nerf_helpers.py:143-205) with a bounding-box affine: the reference's own NeRF and run_network,
    forward, and what its autograd gives for d / d embedded_cam and for the view layer's weight (whose last four columns
    multiply the code) under a fixed cotangent.  Weights: the closed-form depth recipe, the four extra view-layer
    columns from a closed-form sine as well (stored: they are not part of the recipe)."""
    D, DH = import_depth_reference()
    n_cam = 4
    embed_fn, input_ch = DH.get_embedder(9, 0)
    embeddirs_fn, input_ch_views = DH.get_embedder(0, 0)
    net = DH.NeRF(D=8, W=256, input_ch=input_ch, output_ch=5, skips=[4], input_ch_views=input_ch_views,
                  input_ch_cam=n_cam, use_viewdirs=True)
    sd = orc.closed_form_state_dict_depth(3, sharpen=False)
    idx = torch.arange(128 * n_cam, dtype=torch.float64).reshape(128, n_cam)
    extra = (0.05 * torch.sin(0.37 * idx + 1.3)).float()
    sd["views_linears.0.weight"] = torch.cat([sd["views_linears.0.weight"], extra], 1)
    net.load_state_dict(sd)
    gen = torch.Generator().manual_seed(83)
    R, S = 12, 20
    pts = (torch.rand(R, S, 3, generator=gen) * 2 - 1) * 1.5
    vd = torch.nn.functional.normalize(torch.randn(R, 3, generator=gen), dim=-1)
    cam = (torch.randn(n_cam, generator=gen) * 0.3).requires_grad_(True)
    cot = torch.randn(R, S, 4, generator=gen)
    bb_center, bb_scale = 0.15, 0.8
    raw = D.run_network(pts, vd, cam, net, embed_fn=embed_fn, embeddirs_fn=embeddirs_fn, bb_center=bb_center,
                        bb_scale=bb_scale, netchunk=65536)
    (raw * cot).sum().backward()
    npz("g8c_camera_code", pts=pts, viewdirs=vd, cam=cam.detach(), cotangent=cot, bb_center=bb_center, bb_scale=bb_scale,
        view_weight_extra=extra, raw=raw.detach(), grad_cam=cam.grad, grad_view_weight=net.views_linears[0].weight.grad,
        grad_view_bias=net.views_linears[0].bias.grad)


# ---------------------------------------------------------------- G9: checkpoint wire format
def g9():
    """The reference's checkpoint (run_plnerf.py:1324-1332) written after G6 case 0's optimisation step, re-loaded by
    the reference's own create_nerf (run_plnerf.py:454-471), then: a render on G5-style rays with the re-loaded
    networks and one more optimisation step from the re-loaded optimizer state.  The .tar is committed next to the
    fixture: it is what a user of the reference has on disk."""
    import shutil
    args = make_args(64, 128, "linear")
    kw, _, opt, opt_c = ref_nets(args, 0, 1, sharpen=False)
    batch, target = scene_rays(24, seed=60)
    rays = (batch[:, 0:3], batch[:, 3:6])
    Kb = [[1111.111, 0, 400], [0, 1111.111, 400], [0, 0, 1]]

    def step(kw, opt, opt_c):
        rgb, disp, acc, extras = R_.render(800, 800, Kb, chunk=32768, rays=rays, near=2.0, far=6.0, retraw=True,
                                           pytest=True, **kw)
        opt.zero_grad()
        opt_c.zero_grad()
        loss = H_.img2mse(rgb, target) + H_.img2mse(extras["rgb0"], target)
        loss.backward()
        grads = {(tag, name): prm.grad.clone() for net, tag in ((kw["network_fn"], "coarse"), (kw["network_fine"], "fine"))
                 for name, prm in net.named_parameters()}
        opt.step()
        opt_c.step()
        return loss.detach(), grads
    step(kw, opt, opt_c)
    global_step = 1
    path = os.path.join(args.ckpt_dir, args.expname, '{:06d}.tar'.format(global_step))
    torch.save({      # the dict of run_plnerf.py:1326-1331, from the reference's own objects
        'global_step': global_step,
        'network_fn_state_dict': kw['network_fn'].state_dict(),
        'network_fine_state_dict': kw['network_fine'].state_dict(),
        'optimizer_state_dict': opt.state_dict(),
    }, path)
    shutil.copyfile(path, os.path.join(HERE, "g9_reference_checkpoint.tar"))
    # the reference re-loads it
    args2 = Namespace(**dict(vars(args), no_reload=False))
    kw2, _, start2, _, opt2, opt_c2 = R_.create_nerf(args2)
    assert start2 == global_step
    out = {"global_step": start2, "ray_batch": batch, "target": target, "N_samples": 64, "N_importance": 128,
           "sample_stride": 97}
    rbatch, _ = scene_rays(12, seed=90)
    rr = (rbatch[:, 0:3], rbatch[:, 3:6])
    with torch.no_grad():
        rgb, disp, acc, extras = R_.render(800, 800, Kb, chunk=32768, rays=rr, near=2.0, far=6.0, retraw=True,
                                           pytest=True, **kw2)
    out.update({"render_batch": rbatch, "render_rgb_map": rgb, "render_disp_map": disp, "render_acc_map": acc})
    for k, v in extras.items():
        out["render_" + k] = v
    loss2, grads2 = step(kw2, opt2, opt_c2)
    out["loss"] = loss2
    for (tag, name), gr in grads2.items():
        out[f"grad_{tag}_{name}_norm"] = gr.norm()
        out[f"grad_{tag}_{name}_sample"] = sample_elems(gr)
    for net, tag in ((kw2["network_fn"], "coarse"), (kw2["network_fine"], "fine")):
        for name, prm in net.named_parameters():
            out[f"param_{tag}_{name}_sample"] = sample_elems(prm)
    npz("g9_checkpoint", **out)
    print(f"wrote g9_reference_checkpoint.tar ({os.path.getsize(os.path.join(HERE, 'g9_reference_checkpoint.tar'))/1e6:.1f} MB)")


def g8():
    """The depth-supervised variant: pi-scaled encoder + softplus density network
    (57 | 3 input channels), render_rays with the attached `pred_hyp`, space-carving loss, and one clipped
    Adam step over both networks -- all from the reference's own functions."""
    D, DH = import_depth_reference()
    out = {}
    embed_fn, input_ch = DH.get_embedder(9, 0)
    embeddirs_fn, input_ch_views = DH.get_embedder(0, 0)
    assert (input_ch, input_ch_views) == (57, 3)
    coarse, fine = _depth_nets(DH, input_ch, input_ch_views)

    def query(inputs, viewdirs, embedded_cam, fn):
        return D.run_network(inputs, viewdirs, embedded_cam, fn, embed_fn=embed_fn, embeddirs_fn=embeddirs_fn,
                             bb_center=0.0, bb_scale=1.0, netchunk=65536)
    # (a) the network on its own
    gen = torch.Generator().manual_seed(8)
    pts = (torch.rand(6, 16, 3, generator=gen) * 2 - 1) * 2.5
    vd = torch.nn.functional.normalize(torch.randn(6, 3, generator=gen), dim=-1)
    with torch.no_grad():
        raw = query(pts, vd, torch.tensor(()), coarse)
        emb = torch.cat([embed_fn(pts.reshape(-1, 3)), embeddirs_fn(vd[:, None].expand(pts.shape).reshape(-1, 3))], -1)
    out.update({"mlp_pts": pts, "mlp_viewdirs": vd, "mlp_embedded": emb, "mlp_raw": raw})
    # (b) render_rays + (c) one training step
    out.update(_depth_render_and_step(D, DH, coarse, fine, query, R=24, Ns=32, Ni=48, seed=8))
    # ray helpers of the depth script (pixel centres, flipped rows)
    Hh, Ww = 4, 6
    intr = torch.tensor([7.5, 7.25, Ww / 2 - 0.25, Hh / 2 + 0.5])
    c2w_h = orc.pose_spherical(37.0, -30.0, 4.0)[:3, :4]
    ro, rd = DH.get_rays(Hh, Ww, intr, c2w_h)
    coords = torch.tensor([[0, 1], [3, 5], [2, 2]])
    ro_c, rd_c = DH.get_rays(Hh, Ww, intr, c2w_h, coords)
    out.update({"rays_H": Hh, "rays_W": Ww, "rays_intrinsic": intr, "rays_c2w": c2w_h, "rays_o": ro, "rays_d": rd,
                "rays_coords": coords, "rays_o_coords": ro_c, "rays_d_coords": rd_c})
    npz("g8_depth_variant", **out)


if __name__ == "__main__":
    which = sys.argv[1:] or ["g1", "g2", "g3", "g4", "g5", "g6", "g7", "g8", "g8b", "g8c", "g9"]
    for name in which:
        globals()[name]()
