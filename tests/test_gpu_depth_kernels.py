"""Round 3's kernels around the depth-supervised variant of the path, on a real MI355X:

  * plnerf_embed_rows  -- run_network's input assembly (depth_supervised_exps/run_nerf_sample_based_depth.py:52-68 with the
                          pi-scaled Embedder of model/run_nerf_helpers.py:100-130) against the oracle's torch expressions;
  * plnerf_depth_loss  -- the loop's loss (:1126-1150, model/run_nerf_helpers.py:52-86) and its three gradients against
                          torch autograd;
  * the camera code's gradient (input_ch_cam > 0: :1091-1093, 1122-1123, 311-345) through functional.MlpFn against the
    oracle network's autograd in fp64.

Tolerances: the encoding 5e-7 absolute (values in [-1, 1]; sin / cos arguments reach 2^8 pi |x|, one ulp of the result
either side); the loss 1e-6 relative and its gradients 1e-7 absolute + 1e-5 relative (fp64 partial sums in the kernel,
fp32 in torch); the camera gradient 2e-4 of max|g| in fp32 mode and 6e-3 in f16x3 (the half-plane backward, DESIGN.md
section 3).
"""
import numpy as np
import pytest
import torch

from oracle import plnerf_oracle as orc
from test_gpu_parity import assert_close, dev, g

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def P():
    import plnerf_amd
    return plnerf_amd


# (the last case: the widest row the entry point admits -- 262 channels, 67 KB of LDS per workgroup, beyond the 64 KB a
# launch gets without hipFuncAttributeMaxDynamicSharedMemorySize; a float's sin / cos at 2^15 x is still exact enough)
@pytest.mark.parametrize("cfg", [(9, 0, np.pi, 0), (9, 0, np.pi, 4), (10, 4, 1.0, 0), (3, 2, np.pi, 2), (0, 0, 1.0, 0),
                                 (16, 16, 1.0, 64)])
def test_embed_rows_matches_the_reference_expressions(P, cfg):
    from plnerf_amd import functional as Fn
    fx, fd, scale, n_cam = cfg
    gen = torch.Generator().manual_seed(5)
    R, S = 37, 11                                       # ragged against the kernel's 64-row tiles
    pts = (torch.rand(R, S, 3, generator=gen) * 2 - 1) * 3.0
    vd = torch.nn.functional.normalize(torch.randn(R, 3, generator=gen), dim=-1)
    cam = torch.randn(n_cam, generator=gen) if n_cam else None
    center, bscale = torch.tensor([0.3, -0.2, 0.1]), 0.7

    def gamma(x, n):     # model/run_nerf_helpers.py:123: fn(x * pi * freq), in fp32
        blocks = [x]
        for k in range(n):
            xs = x * np.float32(scale) * float(2 ** k)
            blocks += [torch.sin(xs), torch.cos(xs)]
        return torch.cat(blocks, -1)
    flat = (pts.reshape(-1, 3) - center) * bscale
    cols = [gamma(flat, fx), gamma(vd, fd)[:, None, :].expand(R, S, 3 + 6 * fd).reshape(R * S, -1)]
    if n_cam:
        cols.append(cam.reshape(1, -1).expand(R * S, n_cam))
    want = torch.cat(cols, -1)
    got = Fn.embed_rows(g(pts), g(vd), None if cam is None else g(cam), fx, fd, input_scale=scale, bb_center=center,
                        bb_scale=bscale).cpu()
    assert got.shape == want.shape
    err = float((got - want).abs().max())
    print(f"embed_rows fx={fx} fd={fd} scale={scale:.3f} cam={n_cam}: max err {err:.2e}")
    assert err <= (5e-7 if fx <= 10 else 2e-2)          # (band 2^15: the ARGUMENT's own rounding is ~2e-3 rad there)
    assert torch.equal(got[:, :3], want[:, :3])         # the affine and the identity block: bit-equal
    # positions only (no view directions)
    got_x = Fn.embed_rows(g(pts), None, None, fx, 0, input_scale=scale, bb_center=center, bb_scale=bscale).cpu()
    assert got_x.shape[1] == 3 + 6 * fx and float((got_x - want[:, :3 + 6 * fx]).abs().max()) <= (5e-7 if fx <= 10 else 2e-2)


@pytest.mark.parametrize("case", ["full", "no_coarse", "target_per_point", "mask_threshold", "no_depth_term",
                                  "joint", "joint_target_per_point", "joint_mask_threshold"])
def test_depth_loss_and_gradients_match_autograd(P, case):
    """plnerf_depth_loss against autograd of the oracle's restatement of the reference loss, per-ray and is_joint
    (model/run_nerf_helpers.py:52-86); and depth.compute_space_carving_loss -- the reference-named function, backed by
    the same kernel -- as a differentiable scalar."""
    from plnerf_amd import depth as Dp, functional as Fn
    joint = case.startswith("joint")
    gen = torch.Generator().manual_seed(9)
    R, Pn, H = 300, 64, 3
    rgb = torch.rand(R, 3, generator=gen).requires_grad_(True)
    rgb0 = torch.rand(R, 3, generator=gen).requires_grad_(True)
    target = torch.rand(R, 3, generator=gen)
    hyp = (2.0 + 4.0 * torch.rand(R, Pn, generator=gen)).requires_grad_(True)
    th = 2.0 + 4.0 * torch.rand(H, R, Pn if case.endswith("target_per_point") else 1, generator=gen)
    th[1, 5] = hyp.detach()[5, :1] if th.shape[-1] == 1 else hyp.detach()[5]      # an exact hit: |x| at 0
    mask = (torch.rand(R, generator=gen) > 0.3).float() if case.endswith("mask_threshold") else None
    thr = 0.25 if case.endswith("mask_threshold") else 0.0
    w = 0.007
    use0, use_h = case != "no_coarse", case != "no_depth_term"
    loss = torch.mean((rgb - target) ** 2)
    img = loss
    sc = torch.zeros(())
    if use_h:
        sc = orc.compute_space_carving_loss(hyp, th, is_joint=joint, mask=mask, norm_p=2, threshold=thr)
        loss = loss + w * sc
    if use0:
        loss = loss + torch.mean((rgb0 - target) ** 2)
    loss.backward()
    loss5, g1, g0, gh = Fn.depth_loss_and_grads(g(rgb.detach()), g(rgb0.detach()) if use0 else None, g(target),
                                                g(hyp.detach()) if use_h else None, g(th) if use_h else None, w,
                                                threshold=thr, mask=None if mask is None else g(mask), is_joint=joint)
    l5 = loss5.cpu()
    assert abs(float(l5[0]) - float(loss)) <= 1e-6 * abs(float(loss)) + 1e-8, (float(l5[0]), float(loss))
    assert abs(float(l5[1]) - float(img)) <= 1e-6 * float(img) + 1e-8
    assert abs(float(l5[3]) - float(sc)) <= 1e-6 * abs(float(sc)) + 1e-8, (float(l5[3]), float(sc))
    assert abs(float(l5[4]) - float(-10.0 * torch.log10(img.detach()))) <= 1e-4

    def close(a, b, what):
        err = float((a.cpu() - b).abs().max())
        assert err <= 1e-7 + 1e-5 * float(b.abs().max()), f"{case} {what}: {err:.3e}"
    close(g1, rgb.grad, "g_rgb")
    if use0:
        close(g0, rgb0.grad, "g_rgb0")
    if use_h:
        close(gh, hyp.grad, "g_hyp")
        assert float(gh.cpu()[5].abs().max()) >= 0.0      # (finite at the exact hit: torch.norm's sub-gradient 0)
        assert torch.isfinite(gh).all()
        # the reference-named function on the device: same value, and a gradient through autograd
        x = g(hyp.detach()).requires_grad_(True)
        val = Dp.compute_space_carving_loss(x, g(th), is_joint=joint, mask=None if mask is None else g(mask), norm_p=2,
                                            threshold=thr)
        assert abs(float(val) - float(sc)) <= 1e-6 * abs(float(sc)) + 1e-8
        (3.0 * val).backward()
        close(x.grad * (w / 3.0), hyp.grad, "compute_space_carving_loss grad")


@pytest.mark.parametrize("precision", ["fp32", "f16x3"])
def test_camera_code_gradient_through_the_mlp(P, precision):
    """input_ch_cam = 4: the camera code repeated on every row gets d loss / d cam = W_view[:, cam columns]^T applied to
    the row-sum of dz_view; against fp64 autograd of the oracle's network on the same rows."""
    from plnerf_amd import depth as Dp, functional as Fn
    n_cam = 4
    gen = torch.Generator().manual_seed(21)
    R, S = 24, 40
    sd = orc.closed_form_state_dict_depth(3, False)
    extra = torch.randn(128, n_cam, generator=gen) * 0.05
    sd["views_linears.0.weight"] = torch.cat([sd["views_linears.0.weight"], extra], 1)       # [128, 256 + 3 + 4]
    net = P.NeRF(D=8, W=256, input_ch=57, input_ch_views=3, input_ch_cam=n_cam, output_ch=5, skips=[4], use_viewdirs=True,
                 precision=precision, density_activation="softplus")
    net.load_state_dict(sd)
    net = net.to(dev())
    pts = (torch.rand(R, S, 3, generator=gen) * 2 - 1) * 1.5
    vd = torch.nn.functional.normalize(torch.randn(R, 3, generator=gen), dim=-1)
    cam0 = torch.randn(n_cam, generator=gen) * 0.3
    cot = torch.randn(R, S, 4, generator=gen)
    emb_fn, _ = Dp.get_embedder(9, 0)
    embd_fn, _ = Dp.get_embedder(0, 0)
    # fp64 oracle
    cam64 = cam0.double().requires_grad_(True)
    sd64 = {k: v.double().requires_grad_(True) for k, v in sd.items()}
    flat = pts.reshape(-1, 3).double()
    emb = torch.cat([orc.positional_encoding_pi(flat, 9), vd[:, None, :].expand(R, S, 3).reshape(-1, 3).double(),
                     cam64.reshape(1, -1).expand(R * S, n_cam)], -1)
    raw64 = orc.nerf_mlp_depth(sd64, emb).reshape(R, S, 4)
    (raw64 * cot.double()).sum().backward()
    # HIP path through the depth script's run_network
    cam = g(cam0).requires_grad_(True)
    raw = Dp.run_network(g(pts), g(vd), cam, net, emb_fn, embd_fn, 0.0, 1.0)
    err_fwd = float((raw.detach().cpu().double() - raw64.detach()).abs().max())
    (raw * g(cot)).sum().backward()
    gc, gr = cam.grad.cpu().double(), cam64.grad
    rel = float((gc - gr).abs().max()) / float(gr.abs().max())
    gw = net.views_linears[0].weight.grad.cpu().double()
    relw = float((gw - sd64["views_linears.0.weight"].grad).abs().max()) / float(sd64["views_linears.0.weight"].grad.abs().max())
    print(f"{precision}: forward err {err_fwd:.2e}, d/d cam rel err {rel:.2e} (grad {gr.tolist()}), view weight grad rel err {relw:.2e}")
    assert err_fwd <= 2e-5
    assert rel <= (2e-4 if precision == "fp32" else 6e-3)
    # test-time optimisation of the code alone (run_nerf_sample_based_depth.py:311-345): frozen network parameters
    for p_ in net.parameters():
        p_.requires_grad_(False)
    cam2 = g(cam0).requires_grad_(True)
    raw2 = Dp.run_network(g(pts), g(vd), cam2, net, emb_fn, embd_fn, 0.0, 1.0)
    (raw2 * g(cot)).sum().backward()
    rel2 = float((cam2.grad.cpu().double() - gr).abs().max()) / float(gr.abs().max())
    assert rel2 <= (2e-4 if precision == "fp32" else 6e-3)
    # rows that require grad get one (plnerf_mlp_input_grad; checked against the oracle in test_gpu_modes.py)
    x = torch.zeros(4, 64, device=dev(), requires_grad=True)
    net(x).sum().backward()
    assert x.grad is not None and x.grad.shape == (4, 64) and torch.isfinite(x.grad).all() and float(x.grad.abs().max()) > 0


@pytest.mark.parametrize("precision", ["fp32", "f16x3"])
def test_in_kernel_scaled_encoding_matches_the_embedded_route_and_the_oracle(P, precision):
    """The depth variant's pi-scaled encoding evaluated in the MLP kernel's own prologue (NeRF.query(input_scale=pi), what
    depth.run_network does when there is no camera code and no bounding-box affine) against (i) the oracle's network on
    the reference's torch encoding and (ii) the embedded route (plnerf_embed_rows + NeRF.forward) through the same
    weights, forward and parameter gradients; and the NVS encoder with fewer frequencies (multires 6 / multires_views 2:
    39 | 15 channels, a prefix of the compiled 63 | 27) through render.run_network."""
    from plnerf_amd import depth as Dp, functional as Fn
    gen = torch.Generator().manual_seed(31)
    R, S = 21, 50
    sd = orc.closed_form_state_dict_depth(2, True)
    net = P.NeRF(D=8, W=256, input_ch=57, input_ch_views=3, output_ch=5, skips=[4], use_viewdirs=True, precision=precision,
                 density_activation="softplus")
    net.load_state_dict(sd)
    net = net.to(dev())
    assert net.has_fused_encoding()
    pts = (torch.rand(R, S, 3, generator=gen) * 2 - 1) * 1.5
    vd = torch.nn.functional.normalize(torch.randn(R, 3, generator=gen), dim=-1)
    cot = torch.randn(R, S, 4, generator=gen)
    ref = orc.query_network_depth(sd, pts, vd)
    emb_fn, _ = Dp.get_embedder(9, 0)
    embd_fn, _ = Dp.get_embedder(0, 0)
    raw = Dp.run_network(g(pts), g(vd), torch.tensor((), device=dev()), net, emb_fn, embd_fn, 0.0, 1.0)   # fused route
    (raw * g(cot)).sum().backward()
    g_fused = [p.grad.detach().clone() for p in net.parameters()]
    for p_ in net.parameters():
        p_.grad = None
    emb = Fn.embed_rows(g(pts), g(vd), None, 9, 0, input_scale=np.pi)
    raw_e = net(emb).reshape(R, S, 4)
    (raw_e * g(cot)).sum().backward()
    err_o = float((raw.detach().cpu() - ref).abs().max())
    err_e = float((raw.detach() - raw_e.detach()).abs().max())
    worst = max(float((a - p_.grad).abs().max()) / max(float(p_.grad.abs().max()), 1e-9) for a, p_ in zip(g_fused, net.parameters()))
    print(f"{precision}: in-kernel pi encoding vs oracle {err_o:.2e}, vs embedded route {err_e:.2e}, gradients between the routes {worst:.2e}")
    assert err_o <= 1e-5 * (1.0 + float(ref.abs().max())) and err_e <= 2e-6 * (1.0 + float(ref.abs().max()))
    assert worst <= (2e-5 if precision == "fp32" else 2e-3)
    # a bounding-box affine or a camera code keeps the embedded route (same numbers, no refusal)
    raw_b = Dp.run_network(g(pts), g(vd), torch.tensor((), device=dev()), net, emb_fn, embd_fn, 0.1, 0.9)
    ref_b = orc.query_network_depth(sd, pts, vd, bb_center=0.1, bb_scale=0.9)
    assert float((raw_b.detach().cpu() - ref_b).abs().max()) <= 1e-5 * (1.0 + float(ref_b.abs().max()))
    # the NVS encoder with fewer frequencies
    net2 = P.NeRF(D=8, W=256, input_ch=39, input_ch_views=15, output_ch=5, skips=[4], use_viewdirs=True,
                  precision=precision).to(dev())
    e6, _ = P.get_embedder(6, 0)
    e2, _ = P.get_embedder(2, 0)
    with torch.no_grad():
        fused = P.run_network(g(pts), g(vd), net2, e6, e2)
        flat = torch.cat([e6(g(pts).reshape(-1, 3)), e2(g(vd)[:, None].expand(R, S, 3).reshape(-1, 3))], -1)
        generic = net2(flat).reshape(R, S, -1)
    err2 = float((fused - generic).abs().max())
    print(f"{precision}: 39|15-channel network, fused vs torch-embedded {err2:.2e}")
    assert err2 <= 5e-6 * (1.0 + float(generic.abs().max()))


@pytest.mark.parametrize("precision", ["fp32", "f16x3"])
def test_camera_code_against_the_reference_fixture(P, golden, precision):
    """Fixture G8c -- the reference's own NeRF(input_ch_cam = 4) + run_network with a bounding-box affine: raw, d / d
    embedded_cam and the view layer's gradients -- through depth.run_network (plnerf_embed_rows + the EMB kernels +
    MlpFn's camera gradient)."""
    from plnerf_amd import depth as Dp
    gd = golden("g8c_camera_code")
    T = torch.from_numpy
    sd = orc.closed_form_state_dict_depth(3, False)
    sd["views_linears.0.weight"] = torch.cat([sd["views_linears.0.weight"], T(gd["view_weight_extra"])], 1)
    net = P.NeRF(D=8, W=256, input_ch=57, input_ch_views=3, input_ch_cam=4, output_ch=5, skips=[4], use_viewdirs=True,
                 precision=precision, density_activation="softplus")
    net.load_state_dict(sd)
    net = net.to(dev())
    emb_fn, _ = Dp.get_embedder(9, 0)
    embd_fn, _ = Dp.get_embedder(0, 0)
    cam = g(T(gd["cam"])).requires_grad_(True)
    raw = Dp.run_network(g(T(gd["pts"])), g(T(gd["viewdirs"])), cam, net, emb_fn, embd_fn, float(gd["bb_center"]),
                         float(gd["bb_scale"]))
    (raw * g(T(gd["cotangent"]))).sum().backward()
    err = float((raw.detach().cpu() - T(gd["raw"])).abs().max())
    tol = 2e-4 if precision == "fp32" else 6e-3
    rel = {}
    for got, key in ((cam.grad, "grad_cam"), (net.views_linears[0].weight.grad, "grad_view_weight"),
                     (net.views_linears[0].bias.grad, "grad_view_bias")):
        ref = T(gd[key])
        rel[key] = float((got.cpu() - ref).abs().max()) / float(ref.abs().max())
    print(f"{precision} G8c: raw {err:.2e}; " + ", ".join(f"{k} {v:.2e}" for k, v in rel.items()))
    assert err <= 1e-5 * (1.0 + float(T(gd["raw"]).abs().max()))
    assert all(v <= tol for v in rel.values()), rel


@pytest.mark.parametrize("precision", ["fp32", "f16x3", "f16"])
def test_density_activation_inside_the_kernels_equals_softplus_behind_them(P, precision):
    """NeRF(density_activation="softplus") applies F.softplus(sigma, beta=10)
    (depth_supervised_exps/model/run_nerf_helpers.py:200) in the forward kernel's last store and its derivative on the
    backward's entry (plnerf_mlp_fwd / _bwd `density_beta`).  Against the same network without activation followed by
    torch's softplus and autograd: outputs to 1e-6, every parameter gradient to 2e-6 of its largest entry -- the two
    backward passes run the SAME kernels on the same saved state; only where sigmoid(beta sigma) is multiplied in
    differs.  Inputs cover both branches of softplus (beta sigma > 20) and very negative densities."""
    import torch.nn.functional as F
    sd = orc.closed_form_state_dict(0, True)
    sd["alpha_linear.weight"] = sd["alpha_linear.weight"] * 25.0      # densities from far below zero to beta sigma > 20

    def net(act):
        n = P.NeRF(D=8, W=256, input_ch=63, input_ch_views=27, output_ch=5, skips=[4], use_viewdirs=True,
                   precision=precision, density_activation=act)
        n.load_state_dict(sd)
        return n.to(dev())
    a, b = net("softplus"), net(None)
    gen = torch.Generator().manual_seed(3)
    R, S = 70, 33
    pts = g((torch.rand(R, S, 3, generator=gen) * 2 - 1) * 2.5)
    vd = g(torch.nn.functional.normalize(torch.randn(R, 3, generator=gen), dim=-1))
    cot = g(torch.randn(R, S, 4, generator=gen))
    ya = a.query(pts, vd)
    yb = b.query(pts, vd)
    yb = torch.cat([yb[..., :3], F.softplus(yb[..., 3:], beta=10)], -1)
    sig = yb[..., 3]
    assert float(sig.max()) > 2.1 and float(sig.min()) < 1e-6, (float(sig.min()), float(sig.max()))   # both regimes present
    assert_close(ya, yb.detach(), atol=1e-6, rtol=1e-6, what=f"{precision} softplus forward")
    (ya * cot).sum().backward()
    (yb * cot).sum().backward()
    worst = 0.0
    for (name, p), q in zip(a.named_parameters(), b.parameters()):
        scale = max(float(q.grad.abs().max()), 1e-12)
        worst = max(worst, float((p.grad - q.grad).abs().max()) / scale)
    print(f"{precision}: in-kernel softplus vs torch softplus behind the kernel: worst grad diff / max|g| = {worst:.2e}")
    assert worst <= 2e-6 if precision == "fp32" else worst <= 2e-3


def test_depth_render_one_launch_stages_equal_the_separate_launches(P, golden):
    """depth.render_rays in piecewise-linear mode runs the coarse -> fine transition as plnerf_coarse_epilogue and the
    last stage (raw2outputs + the hypotheses' sampler + z_std) as plnerf_fine_epilogue.  Switched off
    (depth.FUSE_STAGES), the same call runs plnerf_quad_fwd / plnerf_sample_pl / plnerf_merge_sort / plnerf_ray_points /
    torch.std one after the other: every output must be bit-identical, and so must the gradients of a loss through the
    maps AND pred_hyp (the fused backward = plnerf_sample_pl_bwd + plnerf_quad_bwd, like the separate one)."""
    from test_gpu_modes import _depth_setup
    from plnerf_amd import depth as Dp
    gd = golden("g8b_depth_variant_128_64")
    batch, _ = orc.synthetic_blender_rays(300, seed=21)
    batch = g(batch)
    outs, grads = [], []
    for fuse in (True, False):
        _, kw, grad_vars, _ = _depth_setup(gd, "fp32")
        Dp.FUSE_STAGES = fuse
        try:
            ret = Dp.render_rays(batch, retraw=True, pytest=True, **kw)
            loss = ret["rgb_map"].sum() + 0.3 * ret["rgb0"].mean() + 0.01 * ret["pred_hyp"].square().mean() + \
                ret["depth_map"].mean()
            loss.backward()
        finally:
            Dp.FUSE_STAGES = True
        outs.append({k: v.detach().clone() for k, v in ret.items()})
        grads.append([p.grad.detach().clone() for p in grad_vars])
    for k in outs[0]:
        if k == "z_std":      # (torch.std's Welford pass against the kernel's two-pass fp64 sum: rounding)
            assert_close(outs[0][k], outs[1][k].cpu(), atol=1e-6, rtol=1e-6, what="z_std")
        else:
            assert torch.equal(outs[0][k], outs[1][k]), k
    for a, b in zip(*grads):
        assert torch.equal(a, b)
    # N_importance = 0: the hypotheses come from the one network's own pass
    _, kw, _, _ = _depth_setup(gd, "fp32")
    kw1 = dict(kw, N_importance=0, network_fine=None)
    single = []
    for fuse in (True, False):
        Dp.FUSE_STAGES = fuse
        try:
            with torch.no_grad():
                single.append(Dp.render_rays(batch, retraw=True, pytest=True, **kw1))
        finally:
            Dp.FUSE_STAGES = True
    for k in single[0]:
        assert torch.equal(single[0][k], single[1][k]), k


def test_depth_step_is_invariant_to_sharding_of_the_batch(P, golden):
    """With a DrawSource installed (DepthTrainStep does) every draw of the depth-supervised render -- jitter, importance
    samples, the hypotheses' u, is_joint's one row -- is a function of (seed, step, GLOBAL ray id): a batch rendered in
    one piece equals the same rays rendered as two shards (to the MLP kernels' summation-order rounding), and the
    returned draws are bit-equal."""
    from test_gpu_modes import _depth_setup
    from plnerf_amd import depth as Dp, functional as Fn
    gd = golden("g8b_depth_variant_128_64")
    _, kw, _, _ = _depth_setup(gd, "f16x3")
    batch, _ = orc.synthetic_blender_rays(96, seed=22)
    rays = g(batch)

    def run(rows, id0, **over):
        prev = Fn.set_draw_source(Fn.DrawSource(seed=8, ray_id0=id0, step=4))
        try:
            with torch.no_grad():
                return Dp.render_rays(rows, retraw=True, **dict(kw, **over))
        finally:
            Fn.set_draw_source(prev)
    for over in ({}, {"is_joint": True}):
        whole = run(rays, 0, **over)
        a, b = run(rays[:40], 0, **over), run(rays[40:], 40, **over)
        assert torch.equal(torch.cat([a["u"], b["u"]], 0), whole["u"])
        if over:
            assert torch.equal(whole["u"][0], whole["u"][-1])      # one row for the whole image
        for k in ("rgb_map", "depth_map", "rgb0", "pred_hyp", "z_vals"):
            assert_close(torch.cat([a[k], b[k]], 0), whole[k].cpu(), atol=2e-4 if k == "pred_hyp" else 2e-5, rtol=2e-5,
                         what=f"two shards {k} {over}")
    other = run(rays, 1)
    assert not torch.equal(other["u"], run(rays, 0)["u"])


def test_depth_step_with_merged_backward_equals_autograd_order(P, golden):
    """DepthTrainStep's merged backward (train.backward_merged: autograd from rgb / rgb0 / pred_hyp down to d loss / d raw
    of both networks, then plnerf_mlp_bwd_multi with the density activation's derivative applied per job) against the same
    steps through torch.autograd.backward: losses to fp32 summation order, weights after three clipped Adam steps to an
    Adam step's rounding."""
    from test_gpu_modes import _depth_args, _depth_setup
    gd = golden("g8b_depth_variant_128_64")
    R = 2048
    batch, target = orc.synthetic_blender_rays(R, seed=9)
    target_h = 2.0 + 4.0 * torch.rand(3, R, 1, generator=torch.Generator().manual_seed(9))
    batch, target, target_h = g(batch), g(target), g(target_h)

    def run(merged):
        Dp, kw, grad_vars, opt = _depth_setup(gd, "f16x3")
        step = Dp.DepthTrainStep(_depth_args(gd, "f16x3"), kw, opt, grad_vars, distributed=False, seed=2)
        step.merged_backward = merged
        out = []
        for _ in range(3):
            loss, img_loss, sc, _ = step(batch, target, target_h)
            out.append((float(loss), float(sc)))
        return out, [p.detach().clone() for p in grad_vars]
    l1, p1 = run(True)
    l0, p0 = run(False)
    for (a, sa), (b, sb) in zip(l1, l0):
        # (the space-carving term runs through the sampler's ill-conditioned closed form: after two Adam steps from gradients
        # summed in another split-K order it differs by 2.8e-5 with round 6's row ranges -- inside 1e-5 with round 5's)
        assert abs(a - b) <= 2e-6 * max(1.0, abs(b)) and abs(sa - sb) <= 1e-4 * max(1.0, abs(sb)), (l1, l0)
    worst = max(float((a - b).abs().max()) for a, b in zip(p1, p0))
    print(f"depth step, merged vs autograd-order backward, 3 steps x {R} rays: loss {l1[-1][0]:.7f} / {l0[-1][0]:.7f}, max parameter difference {worst:.2e}")
    # (Adam moves every weight by ~lr = 5e-4 per step whatever its gradient's size: an entry whose gradient is ~0 -- and this
    # loss runs through the sampler's ill-conditioned closed form -- may flip sign under another summation order: 2 lr)
    assert worst <= 1.1e-3, worst


def test_depth_variant_on_a_shape_outside_the_trunk(P, golden):
    """The depth-supervised variant with netwidth 320 (outside the fused kernels' trunk): its run_network (pi-scaled encoding by
    plnerf_embed_rows, softplus density) goes layer by layer (generic.py) -- against the oracle's restatement of
    model/run_nerf_helpers.py:181-205, which reads its widths off the weights -- and one clipped training step runs."""
    import warnings
    from test_gpu_modes import _depth_args
    from plnerf_amd import depth as Dp
    gd = golden("g8b_depth_variant_128_64")
    args = _depth_args(gd, "f16x3")
    args.netwidth = args.netwidth_fine = 320
    with warnings.catch_warnings(record=True) as caught:
        warnings.simplefilter("always")
        kw, _, _, grad_vars, opt = Dp.create_nerf(args, device=dev())
    assert sum("layer by layer" in str(w.message) for w in caught) == 2
    net = kw["network_fn"]
    assert not net.is_supported()
    gen = torch.Generator().manual_seed(77)
    pts = (torch.rand(9, 33, 3, generator=gen) * 2 - 1) * 0.9
    vd = torch.nn.functional.normalize(torch.randn(9, 3, generator=gen), dim=-1)
    with torch.no_grad():
        raw = kw["network_query_fn"](g(pts), g(vd), torch.tensor((), device=dev()), net)
    sd = {k: v.detach().cpu() for k, v in net.state_dict().items()}
    ref = orc.query_network_depth(sd, pts, vd)
    assert_close(raw, ref, what="depth variant, netwidth 320, layer by layer")
    R = 96
    batch, target = orc.synthetic_blender_rays(R, seed=9)
    target_h = 2.0 + 4.0 * torch.rand(3, R, 1, generator=gen)
    step = Dp.DepthTrainStep(args, kw, opt, grad_vars, distributed=False, seed=2)
    before = [p.detach().clone() for p in grad_vars]
    loss, img_loss, sc, _ = step(g(batch), g(target), g(target_h))
    moved = max(float((a - b.detach()).abs().max()) for a, b in zip(before, grad_vars))
    assert torch.isfinite(loss) and torch.isfinite(sc) and 0.0 < moved <= 5.5e-4
