"""The CPU oracle (oracle/plnerf_oracle.py) against the golden vectors that
tests/golden/make_golden.py generated from the reference itself (G1..G7,
SURVEY.md section 8c).  Runs without a GPU."""
import numpy as np
import pytest
import torch

from oracle import plnerf_oracle as orc

torch.set_num_threads(max(1, min(8, torch.get_num_threads())))
T = torch.from_numpy


def close(a, b, atol=1e-6, rtol=1e-6):
    a = a.detach().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
    np.testing.assert_allclose(a, np.asarray(b), atol=atol, rtol=rtol)


def test_g1_mlp(golden):
    g = golden("g1_mlp")
    pts, vd = T(g["pts"]), T(g["viewdirs"])
    for tag, sharp in (("plain", False), ("sharp", True)):
        sd = orc.closed_form_state_dict(0, sharp)
        raw = orc.query_network(sd, pts, vd)
        close(raw, g[f"raw_{tag}"], atol=2e-6, rtol=2e-6)
        close(orc.nerf_mlp(sd, T(g["embedded"])), g[f"raw_from_embedded_{tag}"], atol=2e-6, rtol=2e-6)
    R, S = pts.shape[:2]
    emb = torch.cat([orc.positional_encoding(pts.reshape(-1, 3), 10),
                     orc.positional_encoding(vd[:, None].expand(R, S, 3).reshape(-1, 3), 4)], -1)
    assert torch.equal(emb, T(g["embedded"]))


def test_g2_raw2outputs(golden):
    g = golden("g2_raw2outputs")
    for c in range(int(g["n_cases"])):
        p = f"c{c}_"
        ffix = bool(g[p + "farcolorfix"]) if (p + "farcolorfix") in g.files else False
        std = float(g[p + "noise_std"])
        res = orc.raw2outputs(T(g[p + "raw"]), T(g[p + "z"]), T(g[p + "near"]), T(g[p + "far"]),
                              T(g[p + "rays_d"]), str(g[p + "mode"]), str(g[p + "color_mode"]),
                              raw_noise_std=std, pytest=std > 0, white_bkgd=bool(g[p + "white_bkgd"]),
                              farcolorfix=ffix)
        names = ["rgb_map", "disp_map", "acc_map", "weights", "depth_map", "tau", "T"]
        for nme, v in zip(names, res):
            if v is None:
                assert (p + nme) not in g.files
            else:
                assert torch.equal(v, T(g[p + nme])), (c, nme)


def test_g3_sample_pdf_bit_exact(golden):
    g = golden("g3_sample_pdf")
    for c in range(int(g["n_cases"])):
        bins, w, N = T(g[f"c{c}_bins"]), T(g[f"c{c}_weights"]), int(g[f"c{c}_N"])
        s, inds = orc.sample_pdf(bins, w, N, det=True, pytest=False, return_inds=True)
        assert torch.equal(inds, T(g[f"c{c}_det_inds"]))
        assert torch.equal(s, T(g[f"c{c}_det_samples"]))
        s, inds = orc.sample_pdf(bins, w, N, det=False, pytest=True, return_inds=True)
        assert torch.equal(inds, T(g[f"c{c}_rnd_inds"]))
        assert torch.equal(s, T(g[f"c{c}_rnd_samples"]))


def test_g4_sample_pl(golden):
    g = golden("g4_sample_pl")
    for c in range(int(g["n_cases"])):
        p = f"c{c}_"
        args = [T(g[p + k]) for k in ("z", "weights", "tau", "T", "near", "far")]
        for kw in ({"pytest": True}, {"u": T(g[p + "u"])}):
            s, Tb, taub, binb, inds = orc.sample_pdf_reformulation(*args, int(g[p + "N"]), det=False,
                                                                   return_inds=True, **kw)
            assert torch.equal(inds, T(g[p + "inds"]))
            for a, k in ((s, "samples"), (Tb, "T_below"), (taub, "tau_below"), (binb, "bin_below")):
                assert torch.equal(a, T(g[p + k])), (c, k)


def test_g5_render_rays(golden):
    g = golden("g5_render_rays")
    for c in range(int(g["n_cases"])):
        p = f"c{c}_"
        o, d = T(g[p + "rays_o"]), T(g[p + "rays_d"])
        near, far = float(g[p + "near"]), float(g[p + "far"])
        vd = d / torch.norm(d, dim=-1, keepdim=True)
        if bool(g[p + "ndc"]):
            o, d = orc.ndc_rays(int(g[p + "H"]), int(g[p + "W"]), float(g[p + "focal"]), 1.0, o, d)
        batch = torch.cat([o, d, near * torch.ones_like(d[:, :1]), far * torch.ones_like(d[:, :1]), vd], -1)
        ret = orc.render_rays(batch, orc.closed_form_state_dict(0, True), orc.closed_form_state_dict(1, True),
                              int(g[p + "N_samples"]), str(g[p + "mode"]), "midpoint", retraw=True,
                              perturb=1.0, N_importance=int(g[p + "N_importance"]),
                              white_bkgd=bool(g[p + "white_bkgd"]), raw_noise_std=float(g[p + "raw_noise_std"]),
                              pytest=True)
        assert set(ret) == {"rgb_map", "disp_map", "acc_map", "depth_map", "raw", "rgb0", "disp0", "depth0",
                            "acc0", "z_std"}
        for k, v in ret.items():
            # same torch ops on the same inputs: the sampler is discontinuous, so anything
            # but near-bit agreement upstream would show up as O(1e-2) errors here.
            close(v, g[p + k], atol=5e-6, rtol=5e-6)


def test_g6_train_step(golden):
    g = golden("g6_train_step")
    stride = int(g["sample_stride"])
    for c in range(int(g["n_cases"])):
        p = f"c{c}_"
        sd_c, sd_f = orc.closed_form_state_dict(0, False), orc.closed_form_state_dict(1, False)
        kw = dict(N_samples=int(g[p + "N_samples"]), N_importance=int(g[p + "N_importance"]), mode="linear",
                  color_mode="midpoint", perturb=1.0, white_bkgd=True, raw_noise_std=0.0, pytest=True)
        loss, g_c, g_f = orc.train_step(sd_c, sd_f, T(g[p + "ray_batch"]), T(g[p + "target"]), kw)
        close(loss, g[p + "loss"], atol=1e-6, rtol=1e-6)
        for tag, grads, sd in (("coarse", g_c, sd_c), ("fine", g_f, sd_f)):
            for name, gr in grads.items():
                ref_norm = float(g[p + f"grad_{tag}_{name}_norm"])
                assert abs(float(gr.norm()) - ref_norm) <= 1e-5 * max(ref_norm, 1e-6) + 1e-9, (tag, name)
                close(gr.reshape(-1)[::stride], g[p + f"grad_{tag}_{name}_sample"], atol=1e-7, rtol=2e-4)
                close(sd[name].detach().reshape(-1)[::stride], g[p + f"param_{tag}_{name}_sample"],
                      atol=2e-6, rtol=1e-5)


def test_g8_depth_variant(golden):
    """The depth-supervised restatement (oracle section 8f-1) against the reference's own outputs: network,
    render_rays with the attached pred_hyp, space-carving loss, gradients and the clipped Adam step."""
    g = golden("g8_depth_variant")
    stride = int(g["sample_stride"])
    sd_c, sd_f = orc.closed_form_state_dict_depth(0, True), orc.closed_form_state_dict_depth(1, True)
    raw = orc.query_network_depth(sd_c, T(g["mlp_pts"]), T(g["mlp_viewdirs"]))
    close(raw, g["mlp_raw"], atol=1e-6, rtol=1e-6)
    flat = T(g["mlp_pts"]).reshape(-1, 3)
    emb = torch.cat([orc.positional_encoding_pi(flat, 9),
                     T(g["mlp_viewdirs"])[:, None].expand(6, 16, 3).reshape(-1, 3)], -1)
    assert torch.equal(emb, T(g["mlp_embedded"]))
    kw = dict(N_samples=int(g["N_samples"]), N_importance=int(g["N_importance"]), mode="linear",
              color_mode="midpoint", perturb=1.0, white_bkgd=True, raw_noise_std=0.0, pytest=True)
    with torch.no_grad():
        ret = orc.render_rays_depth(T(g["ray_batch"]), sd_c, sd_f, **kw)
    assert torch.equal(ret["u"], T(g["render_u"]))
    for k in ("rgb_map", "disp_map", "acc_map", "depth_map", "z_vals", "weights", "pred_hyp", "raw", "rgb0", "disp0",
              "acc0", "depth0", "z_vals0", "weights0", "z_std"):
        close(ret[k], g["render_" + k], atol=2e-6, rtol=2e-5)
    loss, sc, g_c, g_f = orc.depth_train_step(sd_c, sd_f, T(g["ray_batch"]), T(g["target"]), T(g["target_h"]), kw,
                                              space_carving_weight=float(g["space_carving_weight"]))
    close(loss, g["loss"], atol=1e-6, rtol=1e-6)
    close(sc, g["space_carving_loss"], atol=1e-6, rtol=1e-6)
    for tag, grads, sd in (("coarse", g_c, sd_c), ("fine", g_f, sd_f)):
        for name, gr in grads.items():
            ref_norm = float(g[f"grad_{tag}_{name}_norm"])
            assert abs(float(gr.norm()) - ref_norm) <= 1e-5 * max(ref_norm, 1e-6) + 1e-9, (tag, name)
            close(gr.reshape(-1)[::stride], g[f"grad_{tag}_{name}_sample"], atol=1e-7, rtol=2e-4)
            close(sd[name].detach().reshape(-1)[::stride], g[f"param_{tag}_{name}_sample"], atol=2e-6, rtol=1e-5)


def test_g8b_depth_variant_config5_sampling(golden):
    """G8's render + step at BASELINE configs[4]'s sampling (N_samples 128, N_importance 64)."""
    g = golden("g8b_depth_variant_128_64")
    assert (int(g["N_samples"]), int(g["N_importance"])) == (128, 64)
    stride = int(g["sample_stride"])
    sd_c, sd_f = orc.closed_form_state_dict_depth(0, True), orc.closed_form_state_dict_depth(1, True)
    kw = dict(N_samples=128, N_importance=64, mode="linear", color_mode="midpoint", perturb=1.0, white_bkgd=True,
              raw_noise_std=0.0, pytest=True)
    with torch.no_grad():
        ret = orc.render_rays_depth(T(g["ray_batch"]), sd_c, sd_f, **kw)
    assert torch.equal(ret["u"], T(g["render_u"]))
    for k in ("rgb_map", "disp_map", "acc_map", "depth_map", "z_vals", "weights", "pred_hyp", "raw", "rgb0", "disp0",
              "acc0", "depth0", "z_vals0", "weights0", "z_std"):
        close(ret[k], g["render_" + k], atol=2e-6, rtol=2e-5)
    loss, sc, g_c, g_f = orc.depth_train_step(sd_c, sd_f, T(g["ray_batch"]), T(g["target"]), T(g["target_h"]), kw,
                                              space_carving_weight=float(g["space_carving_weight"]))
    close(loss, g["loss"], atol=1e-6, rtol=1e-6)
    close(sc, g["space_carving_loss"], atol=1e-6, rtol=1e-6)
    for tag, grads, sd in (("coarse", g_c, sd_c), ("fine", g_f, sd_f)):
        for name, gr in grads.items():
            ref_norm = float(g[f"grad_{tag}_{name}_norm"])
            assert abs(float(gr.norm()) - ref_norm) <= 1e-5 * max(ref_norm, 1e-6) + 1e-9, (tag, name)
            close(gr.reshape(-1)[::stride], g[f"grad_{tag}_{name}_sample"], atol=1e-7, rtol=2e-4)


def test_g9_reference_checkpoint(golden):
    """The reference-written checkpoint (tests/golden/g9_reference_checkpoint.tar, run_plnerf.py:1324-1332): its
    dict layout, and that the oracle -- started from the file's weights and Adam state -- reproduces what the
    reference computed after re-loading it: a render and one more optimisation step."""
    import os
    g = golden("g9_checkpoint")
    ck = torch.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "g9_reference_checkpoint.tar"),
                    map_location="cpu")
    assert set(ck) == {"global_step", "network_fn_state_dict", "network_fine_state_dict", "optimizer_state_dict"}
    assert ck["global_step"] == int(g["global_step"]) == 1
    names = [k for k, _ in orc.param_shapes()]
    assert list(ck["network_fn_state_dict"]) == names and list(ck["network_fine_state_dict"]) == names
    ost = ck["optimizer_state_dict"]
    assert len(ost["state"]) == 24 and set(ost["state"][0]) == {"step", "exp_avg", "exp_avg_sq"}
    assert ost["param_groups"][0]["lr"] == 5e-4 and tuple(ost["param_groups"][0]["betas"]) == (0.9, 0.999)
    sd_c = {k: v.clone() for k, v in ck["network_fn_state_dict"].items()}
    sd_f = {k: v.clone() for k, v in ck["network_fine_state_dict"].items()}
    kw = dict(N_samples=64, N_importance=128, mode="linear", color_mode="midpoint", perturb=1.0, white_bkgd=True,
              raw_noise_std=0.0, pytest=True)
    with torch.no_grad():
        ret = orc.render_rays(T(g["render_batch"]), sd_c, sd_f, retraw=True, **kw)
    for k in ("rgb_map", "disp_map", "acc_map", "depth_map", "rgb0", "acc0", "depth0", "z_std", "raw"):
        close(ret[k], g["render_" + k], atol=5e-6, rtol=2e-5)
    # one more step: the fine optimizer resumes from the file's state, the coarse one restarts (the reference does
    # not store it)
    state = {}
    params_f = [p.requires_grad_(True) for p in sd_f.values()]
    params_c = [p.requires_grad_(True) for p in sd_c.values()]
    state["opt_f"] = torch.optim.Adam(params_f, lr=5e-4, betas=(0.9, 0.999))
    state["opt_c"] = torch.optim.Adam(params_c, lr=5e-4, betas=(0.9, 0.999))
    state["opt_f"].load_state_dict(ost)
    loss, g_c, g_f = orc.train_step(sd_c, sd_f, T(g["ray_batch"]), T(g["target"]), kw, adam_state=state)
    close(loss, g["loss"], atol=1e-6, rtol=1e-6)
    stride = int(g["sample_stride"])
    for tag, grads, sd in (("coarse", g_c, sd_c), ("fine", g_f, sd_f)):
        for name, gr in grads.items():
            close(gr.reshape(-1)[::stride], g[f"grad_{tag}_{name}_sample"], atol=1e-7, rtol=2e-4)
            close(sd[name].detach().reshape(-1)[::stride], g[f"param_{tag}_{name}_sample"], atol=2e-6, rtol=1e-5)


def test_g7_rays(golden):
    g = golden("g7_rays")
    H, W, f = int(g["H"]), int(g["W"]), float(g["focal"])
    K = [[f, 0, W / 2], [0, f, H / 2], [0, 0, 1]]
    o, d = orc.get_rays(H, W, K, T(g["c2w"]))
    assert torch.equal(o, T(g["rays_o"])) and torch.equal(d, T(g["rays_d"]))
    o2, d2 = orc.ndc_rays(H, W, f, 1.0, o, d)
    assert torch.equal(o2, T(g["ndc_o"])) and torch.equal(d2, T(g["ndc_d"]))


def test_sampler_h4_is_defined_at_u_equal_one():
    """the reference raises for det=True in PL mode (u == 1.0 indexes past tau_diff); the
    oracle (and the HIP kernel) clamp instead."""
    raw, z = torch.randn(4, 16, 4), torch.sort(2 + 4 * torch.rand(4, 16), -1)[0]
    near, far, d = torch.full((4, 1), 2.0), torch.full((4, 1), 6.0), torch.randn(4, 3)
    _, _, _, w, _, tau, Tr = orc.raw2outputs(raw, z, near, far, d, "linear", "midpoint")
    s = orc.sample_pdf_reformulation(z, w, tau, Tr, near, far, 8, det=True)[0]
    assert torch.isfinite(s).all() and (s >= 2.0).all() and (s <= 6.0).all()


def test_g8c_camera_code_through_the_depth_network(golden):
    """The oracle's run_network with a camera code and a bounding-box affine against the reference's own NeRF(input_ch_cam
    = 4) + run_network (fixture G8c): forward, and autograd's d / d embedded_cam and view-layer gradients."""
    gd = golden("g8c_camera_code")
    T = torch.from_numpy
    sd = orc.closed_form_state_dict_depth(3, False)
    sd["views_linears.0.weight"] = torch.cat([sd["views_linears.0.weight"], T(gd["view_weight_extra"])], 1)
    sd = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    cam = T(gd["cam"]).clone().requires_grad_(True)
    raw = orc.query_network_depth(sd, T(gd["pts"]), T(gd["viewdirs"]), bb_center=float(gd["bb_center"]),
                                  bb_scale=float(gd["bb_scale"]), embedded_cam=cam)
    (raw * T(gd["cotangent"])).sum().backward()
    assert float((raw.detach() - T(gd["raw"])).abs().max()) <= 2e-6
    for got, key in ((cam.grad, "grad_cam"), (sd["views_linears.0.weight"].grad, "grad_view_weight"),
                     (sd["views_linears.0.bias"].grad, "grad_view_bias")):
        ref = T(gd[key])
        assert float((got - ref).abs().max()) <= 2e-5 * float(ref.abs().max()) + 1e-7, key
    # the identity the HIP path uses: every row carries the same code, so its gradient is the view layer's weight
    # columns applied to the row-sum of dz_view, i.e. to the view layer's bias gradient
    wv = sd["views_linears.0.weight"].detach()
    via_bias = wv[:, -cam.numel():].t() @ T(gd["grad_view_bias"])
    assert float((via_bias - T(gd["grad_cam"])).abs().max()) <= 2e-5 * float(T(gd["grad_cam"]).abs().max())
