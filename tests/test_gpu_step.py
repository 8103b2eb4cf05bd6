"""The step-level kernels (csrc/epilogue.hip, csrc/step.hip) on a real MI355X: the fused coarse epilogue against the
separate launches it replaces (bit for bit), the counter-based draws (known-answer + invariance to how a batch is
split), device-side ray selection against the get_rays formula, the fused image loss against torch, and the training
step built from them.
"""
import os
import tempfile
from argparse import Namespace

import numpy as np
import pytest
import torch

from oracle import plnerf_oracle as orc
from test_gpu_parity import assert_close, dev, g, make_net, maxdiff, quad_case

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def P():
    import plnerf_amd
    return plnerf_amd


# ----------------------------------------------------------------------------- Philox (host restatement for the KAT)
def _philox4x32_10(ctr, key):
    M0, M1, W0, W1 = 0xD2511F53, 0xCD9E8D57, 0x9E3779B9, 0xBB67AE85
    c, k = list(ctr), list(key)
    for _ in range(10):
        p0, p1 = M0 * c[0], M1 * c[2]
        c = [(p1 >> 32) ^ c[1] ^ k[0], p1 & 0xFFFFFFFF, (p0 >> 32) ^ c[3] ^ k[1], p0 & 0xFFFFFFFF]
        k = [(k[0] + W0) & 0xFFFFFFFF, (k[1] + W1) & 0xFFFFFFFF]
    return c


def test_philox_known_answers_and_device_draws(P):
    from plnerf_amd import functional as Fn
    # Random123's known-answer vectors for philox4x32-10
    assert _philox4x32_10([0, 0, 0, 0], [0, 0]) == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    assert _philox4x32_10([0xffffffff] * 4, [0xffffffff] * 2) == [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]
    assert _philox4x32_10([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0]) == \
        [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]
    seed, step, stream_id, id0 = 0x1234567890abcdef, 7, 1, 1000
    src = Fn.DrawSource(seed=seed, ray_id0=id0, step=step)
    out = src.uniform(5, 10, stream_id, dev()).cpu().numpy()
    for r in range(5):
        for col in range(10):
            w = _philox4x32_10([id0 + r, col >> 2, stream_id, step], [seed & 0xFFFFFFFF, seed >> 32])[col & 3]
            assert out[r, col] == np.float32((w >> 8) * 2.0 ** -24), (r, col)
    big = src.uniform(4096, 128, 0, dev())
    assert float(big.min()) >= 0.0 and float(big.max()) < 1.0
    assert abs(float(big.mean()) - 0.5) < 2e-3 and abs(float(big.var()) - 1.0 / 12.0) < 1e-3
    # a batch drawn in one piece == the same global rows drawn as two shards
    a = Fn.DrawSource(seed=3, ray_id0=0, step=2).uniform(8, 64, 0, dev())
    b = torch.cat([Fn.DrawSource(seed=3, ray_id0=0, step=2).uniform(5, 64, 0, dev()),
                   Fn.DrawSource(seed=3, ray_id0=5, step=2).uniform(3, 64, 0, dev())], 0)
    assert torch.equal(a, b)
    assert not torch.equal(a, Fn.DrawSource(seed=3, ray_id0=0, step=3).uniform(8, 64, 0, dev()))
    # normal draws (the density noise): Box-Muller on the same Philox block, restated on the host
    nrm = src.normal(5, 10, 2, dev()).cpu().numpy()
    for r in range(5):
        for col in range(10):
            w = _philox4x32_10([id0 + r, col >> 2, 2, step], [seed & 0xFFFFFFFF, seed >> 32])
            h = (col & 3) >> 1
            u1 = 1.0 - np.float64(np.float32((w[2 * h] >> 8) * 2.0 ** -24))
            u2 = np.float64(np.float32((w[2 * h + 1] >> 8) * 2.0 ** -24))
            rad, ang = np.sqrt(-2.0 * np.log(u1)), 2.0 * np.pi * u2
            want = rad * (np.cos(ang) if (col & 1) == 0 else np.sin(ang))
            assert abs(nrm[r, col] - want) <= 2e-6 * (1.0 + abs(want)), (r, col, nrm[r, col], want)
    bign = src.normal(4096, 192, 2, dev())
    assert torch.isfinite(bign).all() and abs(float(bign.mean())) < 5e-3 and abs(float(bign.var()) - 1.0) < 1e-2
    assert abs(float((bign ** 4).mean()) - 3.0) < 0.1                     # kurtosis of a normal
    n1 = torch.cat([Fn.DrawSource(seed=3, ray_id0=0, step=2).normal(5, 64, 2, dev()),
                    Fn.DrawSource(seed=3, ray_id0=5, step=2).normal(3, 64, 2, dev())], 0)
    assert torch.equal(n1, Fn.DrawSource(seed=3, ray_id0=0, step=2).normal(8, 64, 2, dev()))


# ----------------------------------------------------------------------------- fused coarse epilogue
@pytest.mark.parametrize("S,N,color,white", [(64, 128, "midpoint", True), (128, 64, "midpoint", False),
                                             (37, 23, "left", True), (300, 200, "midpoint", True), (2, 1, "midpoint", False)])
def test_fused_coarse_epilogue_equals_separate_launches(P, S, N, color, white):
    """plnerf_coarse_epilogue == plnerf_quad_fwd -> plnerf_sample_pl -> clamp -> plnerf_merge_sort ->
    plnerf_ray_points, bit for bit (same device functions, same LDS values), forward and backward; z_std against
    torch.std."""
    from plnerf_amd import functional as Fn
    R = 133
    raw, z, near, far, d, _ = quad_case(R, S, 1000 + S)
    gen = torch.Generator().manual_seed(S * 7 + N)
    o = torch.randn(R, 3, generator=gen)
    u = torch.rand(R, N, generator=gen)
    noise = torch.rand(R, S, generator=gen) if S == 37 else None
    raw_a = g(raw).requires_grad_(True)
    rgb, disp, acc, w, depth, tau, Tr = Fn.QuadratureFn.apply(raw_a, g(z), g(near), g(far), g(d), None if noise is None else g(noise),
                                                              "linear", color, white, False)
    zs = Fn.sample_pl(g(z), w, tau, Tr, g(near), g(far), g(u), 1e-4, 1e-3).detach()
    z_fine = Fn.merge_sort(g(z), zs, g(near), g(far))
    pts = Fn.ray_points(g(o), g(d), z_fine)
    z_std = torch.std(torch.clamp(zs, g(near), g(far)), dim=-1, unbiased=False)
    raw_b = g(raw).requires_grad_(True)
    out = Fn.CoarseEpilogueFn.apply(raw_b, g(z), g(near), g(far), g(o), g(d), None if noise is None else g(noise), g(u),
                                    N, color, white, False, 1e-4, 1e-3, None)
    for name, a, b in (("rgb0", rgb, out[0]), ("disp0", disp, out[1]), ("acc0", acc, out[2]), ("depth0", depth, out[3]),
                       ("z_fine", z_fine, out[4]), ("pts", pts, out[5])):
        assert torch.equal(a, b), (name, maxdiff(a, b))
    assert_close(out[6], z_std.cpu(), atol=2e-6, rtol=2e-6, what="z_std")
    cot = torch.randn(R, 3, generator=gen)
    (rgb * g(cot)).sum().backward()
    (out[0] * g(cot)).sum().backward()
    assert torch.equal(raw_a.grad, raw_b.grad)
    # draws made inside the kernel == the same draws handed in as a tensor
    src = Fn.DrawSource(seed=11, ray_id0=40, step=5)
    u_dev = src.uniform(R, N, Fn.DrawSource.U, dev())
    a = Fn.CoarseEpilogueFn.apply(g(raw), g(z), g(near), g(far), g(o), g(d), None, u_dev, N, color, white, False, 1e-4,
                                  1e-3, None)
    b = Fn.CoarseEpilogueFn.apply(g(raw), g(z), g(near), g(far), g(o), g(d), None, None, N, color, white, False, 1e-4,
                                  1e-3, src)
    for x, y in zip(a, b):
        assert torch.equal(x, y)


@pytest.mark.parametrize("S,lindisp", [(64, False), (128, False), (37, True), (1, False)])
def test_coarse_samples_equals_separate_launches(P, S, lindisp):
    from plnerf_amd import functional as Fn
    R = 77
    gen = torch.Generator().manual_seed(S)
    o, d = g(torch.randn(R, 3, generator=gen)), g(torch.randn(R, 3, generator=gen))
    near = g(1.0 + torch.rand(R, 1, generator=gen))
    far = near + 3.0
    t_vals = Fn.cpu_linspace(S, dev())
    src = Fn.DrawSource(seed=5, ray_id0=9, step=1)
    t_rand = src.uniform(R, S, Fn.DrawSource.T_RAND, dev())
    for tr, draws, perturb in ((None, None, False), (t_rand, None, True), (None, src, True)):
        z_ref = Fn.stratified_z(near, far, t_vals, t_rand if perturb else None, lindisp)
        p_ref = Fn.ray_points(o, d, z_ref)
        z, p = Fn.coarse_samples(o, d, near, far, t_vals, tr, lindisp, perturb, draws)
        assert torch.equal(z, z_ref) and torch.equal(p, p_ref), (S, lindisp, perturb)


# ----------------------------------------------------------------------------- ray selection, loss
def test_select_view_rays(P):
    H, W, f = 40, 60, 50.0
    K = [[f, 0, W / 2], [0, f, H / 2], [0, 0, 1]]
    c2w = P.rays.pose_spherical(40.0, -30.0, 4.0)[:3, :4]
    image = torch.rand(H, W, 3, generator=torch.Generator().manual_seed(0))
    o_all, d_all = P.get_rays(H, W, K, c2w)
    for precrop, n in ((None, 500), ((7, 9), 2 * 7 * 2 * 9)):      # the second draw takes EVERY pixel of the window
        cols, target, pix = P.select_view_rays(H, W, K, c2w, g(image), n, 2.0, 6.0, seed=4, step=3, precrop=precrop,
                                               want_pixels=True)
        rows, cs = pix[:, 0].long().cpu(), pix[:, 1].long().cpu()
        assert len(set((rows * W + cs).tolist())) == n                                   # distinct pixels
        if precrop is not None:
            assert rows.min() == H // 2 - 7 and rows.max() == H // 2 + 6 and cs.min() == W // 2 - 9 and cs.max() == W // 2 + 8
        assert_close(cols.rays_d, d_all[rows, cs], atol=1e-6, rtol=1e-6, what="rays_d")
        assert torch.equal(cols.rays_o.cpu(), o_all[rows, cs])
        assert_close(cols.viewdirs, torch.nn.functional.normalize(d_all[rows, cs], dim=-1), atol=1e-6, rtol=1e-6, what="viewdirs")
        assert torch.equal(target.cpu(), image[rows, cs])
        assert (cols.near == 2.0).all() and (cols.far == 6.0).all()
    # two ranks of a global batch of 300 draw the same pixels one rank would, and disjoint ones
    whole = P.select_view_rays(H, W, K, c2w, g(image), 300, 2.0, 6.0, seed=4, step=9, want_pixels=True)
    h0 = P.select_view_rays(H, W, K, c2w, g(image), 150, 2.0, 6.0, seed=4, step=9, ray_id0=0, want_pixels=True)
    h1 = P.select_view_rays(H, W, K, c2w, g(image), 150, 2.0, 6.0, seed=4, step=9, ray_id0=150, want_pixels=True)
    assert torch.equal(whole[2], torch.cat([h0[2], h1[2]], 0)) and torch.equal(whole[1], torch.cat([h0[1], h1[1]], 0))
    other = P.select_view_rays(H, W, K, c2w, g(image), 300, 2.0, 6.0, seed=4, step=10, want_pixels=True)
    assert not torch.equal(whole[2], other[2])
    # the choice is roughly uniform over the image: mean pixel coordinates near the centre
    many = P.select_view_rays(800, 800, [[1111.0, 0, 400], [0, 1111.0, 400], [0, 0, 1]], c2w, None, 65536, 2.0, 6.0,
                              seed=1, step=0, want_pixels=True)[2].float()
    assert abs(float(many[:, 0].mean()) - 399.5) < 4.0 and abs(float(many[:, 1].mean()) - 399.5) < 4.0


def test_ndc_rays_kernel_is_the_reference_expression(P):
    """plnerf_ndc_rays (one launch) against the oracle's ndc_rays (run_nerf_helpers.py:184-201) on the host: the same
    bits, for the LLFF geometry of BASELINE configs[3] and for a focal / near that are not fp32 numbers; rays that need
    a gradient, or live on the host, take the expression."""
    gen = torch.Generator().manual_seed(11)
    for (H, W, focal, near, shape) in ((378, 504, 407.5658, 1.0, (4096,)), (60, 80, 1.0 / 3.0 + 50.0, 0.7, (13, 17))):
        o = torch.randn(*shape, 3, generator=gen) * 0.3
        d = torch.randn(*shape, 3, generator=gen) * 0.4
        d[..., 2] = -(0.5 + torch.rand(*shape, generator=gen))
        ref_o, ref_d = orc.ndc_rays(H, W, focal, near, o, d)
        got_o, got_d = P.rays.ndc_rays(H, W, focal, near, g(o), g(d))
        assert got_o.is_cuda and got_o.shape == o.shape
        assert torch.equal(got_o.cpu(), ref_o) and torch.equal(got_d.cpu(), ref_d)
        host_o, host_d = P.rays.ndc_rays(H, W, focal, near, o, d)                     # host tensors: the expression
        assert torch.equal(host_o, ref_o) and torch.equal(host_d, ref_d)
        og = g(o).requires_grad_(True)
        with_grad, _ = P.rays.ndc_rays(H, W, focal, near, og, g(d))                   # a gradient is asked for: the expression
        assert with_grad.requires_grad and maxdiff(with_grad, ref_o) <= 1e-6
    from plnerf_amd import _lib as L
    assert L.lib().plnerf_ndc_rays(0, 4, 1.0, 1.0, None, None, 4, None, None, None) == -1      # PLNERF_EINVAL
    assert L.lib().plnerf_ndc_rays(4, 4, 1.0, 1.0, None, None, 0, None, None, None) == 0


def test_image_loss_matches_torch(P):
    from plnerf_amd import functional as Fn
    gen = torch.Generator().manual_seed(2)
    R = 4096
    rgb, rgb0, target = (torch.rand(R, 3, generator=gen) for _ in range(3))
    a, b = rgb.clone().requires_grad_(True), rgb0.clone().requires_grad_(True)
    ref = torch.mean((a - target) ** 2) + torch.mean((b - target) ** 2)
    ref.backward()
    x, y = g(rgb).requires_grad_(True), g(rgb0).requires_grad_(True)
    total, fine, coarse = Fn.ImageLossFn.apply(x, y, g(target))
    assert abs(float(total) - float(ref)) <= 1e-7 and abs(float(fine) - float(torch.mean((rgb - target) ** 2))) <= 1e-7
    (3.0 * total + 0.5 * fine).backward()
    assert_close(x.grad, 3.5 * a.grad, atol=1e-9, rtol=1e-6, what="g_rgb")
    assert_close(y.grad, 3.0 * b.grad, atol=1e-9, rtol=1e-6, what="g_rgb0")
    x2 = g(rgb).requires_grad_(True)
    t2, f2, c2 = Fn.ImageLossFn.apply(x2, None, g(target))      # single-pass configuration
    t2.backward()
    assert abs(float(t2) - float(f2)) == 0.0 and float(c2) == 0.0
    assert_close(x2.grad, a.grad, atol=1e-9, rtol=1e-6, what="g_rgb single")


# ----------------------------------------------------------------------------- the step built from them
def _args(ckpt_dir, precision="f16x3", **over):
    a = dict(multires=10, i_embed=0, use_viewdirs=True, multires_views=4, N_importance=128, N_samples=64, netdepth=8,
             netwidth=256, netdepth_fine=8, netwidth_fine=256, netchunk=65536, lrate=5e-4, coarse_lrate=5e-4,
             ft_path=None, ckpt_dir=ckpt_dir, expname="exp", no_reload=True, perturb=1.0, white_bkgd=True,
             raw_noise_std=0.0, mode="linear", color_mode="midpoint", dataset="blender", no_ndc=False, lindisp=False,
             lrate_decay=250, constant_init=0, chunk=32768, precision=precision, N_rand=256)
    a.update(over)
    return Namespace(**a)


def _nets(P, precision="f16x3", **over):
    d = tempfile.mkdtemp()
    os.makedirs(os.path.join(d, "exp"))
    args = _args(d, precision, **over)
    kw, _, start, _, opt, opt_c = P.create_nerf(args, device=dev())
    kw["network_fn"].load_state_dict(orc.closed_form_state_dict(0, False))
    kw["network_fine"].load_state_dict(orc.closed_form_state_dict(1, False))
    return args, kw, opt, opt_c


def test_render_is_invariant_to_sharding_of_the_batch(P):
    """Counter-based draws: rendering a global batch in one piece, as two "ranks" (ray_id0 offsets), or in chunks gives
    the same per-ray results -- what makes a data-parallel step independent of the world size.  The DRAWS are
    bit-identical (test_philox_known_answers_and_device_draws, test_coarse_samples_equals_separate_launches); the
    rendered values agree to rounding only, because the 16-bit MLP kernels rotate the k-step order by workgroup
    (DESIGN.md section 4), so a row's fp32 summation order depends on its position in the launch."""
    from plnerf_amd import functional as Fn
    args, kw, _, _ = _nets(P)
    batch, _ = orc.synthetic_blender_rays(96, seed=21)
    rays = g(batch)
    rkw = {k: v for k, v in kw.items() if k not in ("ndc", "use_viewdirs")}

    def run(rows, id0, chunk=None):
        prev = Fn.set_draw_source(Fn.DrawSource(seed=8, ray_id0=id0, step=4))
        try:
            with torch.no_grad():
                if chunk is None:
                    return P.render_rays(rows, retraw=True, **rkw)
                return P.batchify_rays(rows, chunk, retraw=True, **rkw)
        finally:
            Fn.set_draw_source(prev)
    whole = run(rays, 0)
    a, b = run(rays[:40], 0), run(rays[40:], 40)
    chunked = run(rays, 0, chunk=36)
    for k in ("rgb_map", "depth_map", "rgb0", "z_std"):
        assert_close(torch.cat([a[k], b[k]], 0), whole[k].cpu(), atol=2e-5, rtol=2e-5, what=f"two shards {k}")
        assert_close(chunked[k], whole[k].cpu(), atol=2e-5, rtol=2e-5, what=f"chunked {k}")
    other = run(rays, 1)          # the same rays under other global ids: different draws
    assert maxdiff(whole["depth_map"], other["depth_map"]) > 1e-3
    # the density noise of the LLFF configurations (raw_noise_std = 1: run_plnerf.py:568-570) comes from the same
    # counters: sharding- and chunking-invariant too, different between the coarse and the fine pass, and it matters
    rkw = dict(rkw, raw_noise_std=1.0)
    whole_n = run(rays, 0)
    a, b = run(rays[:40], 0), run(rays[40:], 40)
    chunked = run(rays, 0, chunk=36)
    for k in ("rgb_map", "depth_map", "rgb0", "z_std"):
        assert_close(torch.cat([a[k], b[k]], 0), whole_n[k].cpu(), atol=2e-5, rtol=2e-5, what=f"noise, two shards {k}")
        assert_close(chunked[k], whole_n[k].cpu(), atol=2e-5, rtol=2e-5, what=f"noise, chunked {k}")
    assert maxdiff(whole_n["rgb0"], whole["rgb0"]) > 1e-4


@pytest.mark.parametrize("dataset", ["blender", "llff"])
def test_train_step_from_a_view(P, dataset):
    """TrainStep.step_view (device-side pixel choice -> columns -> render_rays -> fused loss -> backward -> Adam) equals
    TrainStep.__call__ on the same rays and targets, and optimises.  "llff": a forward-facing view in normalised device
    coordinates (create_nerf leaves ndc on, run_plnerf.py:490-493; near 0, far 1, :1008-1009) -- step_view warps the
    selected columns itself, __call__ goes through render's own warp."""
    H = W = 100
    K = [[140.0, 0, W / 2], [0, 140.0, H / 2], [0, 0, 1]]
    if dataset == "llff":
        c2w = torch.tensor([[1.0, 0.0, 0.0, 0.1], [0.0, 1.0, 0.0, -0.05], [0.0, 0.0, 1.0, 0.2]])
        near, far = 0.0, 1.0
    else:
        c2w = P.rays.pose_spherical(30.0, -30.0, 4.0)[:3, :4]
        near, far = 2.0, 6.0
    yy, xx = torch.meshgrid(torch.linspace(0, 1, H), torch.linspace(0, 1, W), indexing="ij")
    image = g(torch.stack([xx, yy, 0.5 * (xx + yy)], -1))
    args, kw, opt, opt_c = _nets(P, dataset=dataset)
    assert bool(kw.get("ndc", True)) == (dataset == "llff")
    ts = P.TrainStep(args, kw, opt, opt_c, distributed=False, seed=5)
    args2, kw2, opt2, opt_c2 = _nets(P, dataset=dataset)
    ts2 = P.TrainStep(args2, kw2, opt2, opt_c2, distributed=False, seed=5)
    losses = []
    for step in range(6):
        cols, target, _ = P.select_view_rays(H, W, K, c2w, image, 256, near, far, seed=5, step=step)
        loss2, psnr2 = ts2(H, W, K, (cols.rays_o, cols.rays_d), target, near=near, far=far)
        loss, psnr = ts.step_view(H, W, K, c2w, image, near=near, far=far, n_rand=256)
        losses.append(float(loss))
        assert abs(float(loss) - float(loss2)) <= 2e-6 * max(1.0, float(loss2)), (step, float(loss), float(loss2))
    assert np.isfinite(losses).all() and losses[-1] < losses[0]
    for p, q in zip(kw["network_fine"].parameters(), kw2["network_fine"].parameters()):
        assert maxdiff(p, q) <= 3e-4      # (six Adam steps apart: the two routes normalise view directions differently)


def test_single_pass_configuration_steps_two_adams_over_one_flat_buffer(P):
    """N_importance = 0: the reference builds BOTH Adams over the coarse network's parameters and steps them one after the
    other (run_plnerf.py:438-447, 1302-1303).  Here both are FlatAdam: the second adopts the flat buffer the first one
    re-homed the weights into (a second re-homing would orphan the first optimizer's buffer), both carry the network's
    range guard, and two steps of the train loop land where two torch.optim.Adam instances over the same tensors land."""
    from plnerf_amd.optim import FlatAdam, flat_view_of
    d = tempfile.mkdtemp()
    os.makedirs(os.path.join(d, "exp"))
    args = _args(d, "fp32", N_importance=0)
    torch.manual_seed(0)
    kw, _, _, grad_vars, opt, opt_c = P.create_nerf(args, device=dev())
    assert kw["network_fine"] is None and isinstance(opt, FlatAdam) and isinstance(opt_c, FlatAdam)
    net = kw["network_fn"]
    assert flat_view_of([p.data for p in net.parameters()]).data_ptr() == opt._flat[0]["param"].data_ptr() \
        == opt_c._flat[0]["param"].data_ptr()
    ref = [p.detach().clone().requires_grad_(True) for p in net.parameters()]
    ref_opts = [torch.optim.Adam(ref, lr=args.lrate, betas=(0.9, 0.999)), torch.optim.Adam(ref, lr=args.coarse_lrate,
                                                                                         betas=(0.9, 0.999))]
    batch, target = orc.synthetic_blender_rays(256, seed=4)
    ts = P.TrainStep(args, kw, opt, opt_c, distributed=False, seed=0)
    K = [[1111.111, 0, 400], [0, 1111.111, 400], [0, 0, 1]]
    for _ in range(2):
        loss, _ = ts(800, 800, K, (g(batch[:, 0:3]), g(batch[:, 3:6])), g(target), near=2.0, far=6.0)
        for r, p in zip(ref, net.parameters()):
            r.grad = p.grad.detach().clone()
        lr = opt.param_groups[0]["lr"]
        for o in ref_opts:
            o.step()
            for grp in o.param_groups:
                grp["lr"] = lr
        assert torch.isfinite(loss)
    worst = max(float((r.detach() - p.detach()).abs().max()) for r, p in zip(ref, net.parameters()))
    print(f"single-pass, two Adams, two steps: max parameter difference to torch.optim.Adam x 2 = {worst:.2e}")
    assert worst <= 2e-6


def test_training_step_is_bitwise_reproducible(P):
    """Two runs of the benchmark-sized step (4096 rays x (64 + 128) samples, f16x3) from the same weights and seed end in
    bit-identical parameters, step after step.  Every reduction of the path is ordered (split-K partials summed in a
    fixed order, no floating-point atomics) and every draw is a counter-based function of (seed, step, ray), so anything
    else -- a weight fragment read before its DMA landed, a plane overwritten early -- would show up here as a
    difference: the test is the race detector for the register-resident kernel's counted vmcnt waits."""
    H = W = 200
    K = [[280.0, 0, W / 2], [0, 280.0, H / 2], [0, 0, 1]]
    gen = torch.Generator().manual_seed(3)
    image = g(torch.rand(H, W, 3, generator=gen))
    poses = [P.rays.pose_spherical(-180.0 + 90.0 * i, -30.0, 4.0)[:3, :4] for i in range(4)]

    def run():
        args, kw, opt, opt_c = _nets(P)
        ts = P.TrainStep(args, kw, opt, opt_c, distributed=False, seed=11)
        digests = []
        for step in range(8):
            loss, _ = ts.step_view(H, W, K, poses[step % 4], image, near=2.0, far=6.0, n_rand=4096)
            digests.append((float(loss), [p.detach().clone() for n in ts.nets for p in n.parameters()]))
        return digests

    a, b = run(), run()
    for step, ((la, pa), (lb, pb)) in enumerate(zip(a, b)):
        assert la == lb, (step, la, lb)
        for x, y in zip(pa, pb):
            assert torch.equal(x, y), f"parameters differ after step {step}"


def test_image_loss_takes_the_coarse_term_from_an_earlier_launch(P):
    """plnerf_image_loss(rgb, NULL, target, coarse_loss = the loss4 of a launch on rgb0 alone) == one launch over both
    images, bit for bit: what lets the coarse network's loss and backward start before the fine pass exists."""
    from plnerf_amd import functional as Fn
    gen = torch.Generator().manual_seed(9)
    rgb, rgb0, target = (g(torch.rand(4096, 3, generator=gen)) for _ in range(3))
    both, g1, g0 = Fn.image_loss_and_grads(rgb, rgb0, target)
    first, gc, _ = Fn.image_loss_and_grads(rgb0, None, target)
    second, gf, none = Fn.image_loss_and_grads(rgb, None, target, coarse_loss=first)
    assert none is None and torch.equal(second, both) and torch.equal(gf, g1) and torch.equal(gc, g0)
    ref = torch.mean((rgb.double() - target.double()) ** 2), torch.mean((rgb0.double() - target.double()) ** 2)
    assert abs(float(both[1]) - float(ref[0])) <= 1e-7 and abs(float(both[2]) - float(ref[1])) <= 1e-7
    assert abs(float(both[3]) + 10.0 * np.log10(float(ref[0]))) <= 1e-4
    for R in (1, 5, 77, 1000):      # fewer elements than workgroups x threads, ragged slices
        a, b, t = (g(torch.rand(R, 3, generator=gen)) for _ in range(3))
        l4, ga, gb = Fn.image_loss_and_grads(a, b, t)
        assert abs(float(l4[0]) - float(torch.mean((a - t) ** 2) + torch.mean((b - t) ** 2))) <= 2e-7
        assert_close(ga, (2.0 / (3 * R)) * (a - t).cpu(), atol=1e-9, rtol=1e-6, what=f"g_rgb R={R}")


# ----------------------------------------------------------------------------- data parallel step, N ranks on one GPU
_DP_GPU_WORKER = r"""
import os, sys, tempfile, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
sys.path.insert(0, os.path.join(sys.argv[1], "tests"))
import plnerf_amd as P
from plnerf_amd import dp
from oracle import plnerf_oracle as orc
from test_gpu_step import _args
HW, N_RAND, STEPS = int(sys.argv[2]), int(sys.argv[3]), 3      # view size, rays PER RANK
rank, world, _ = dp.init_from_env(backend="gloo")          # every rank on cuda:0; gloo moves CUDA tensors through the host
dev = torch.device("cuda:0")
torch.cuda.set_device(0)
FLAGGED = [r for r in (1, 5) if r < world]                  # ranks whose fine forward "leaves the half range" below


def make(distributed):
    d = tempfile.mkdtemp(); os.makedirs(os.path.join(d, "exp"))
    args = _args(d, "f16x3", chunk=32768)
    kw, _, _, _, opt, opt_c = P.create_nerf(args, device=dev)
    kw["network_fn"].load_state_dict(orc.closed_form_state_dict(0, False))
    kw["network_fine"].load_state_dict(orc.closed_form_state_dict(1, False))
    return kw, P.TrainStep(args, kw, opt, opt_c, distributed=distributed, seed=3)


H = W = HW
K = [[1.4 * HW, 0, W / 2], [0, 1.4 * HW, H / 2], [0, 0, 1]]
c2w = P.rays.pose_spherical(20.0, -30.0, 4.0)[:3, :4]
yy, xx = torch.meshgrid(torch.linspace(0, 1, H), torch.linspace(0, 1, W), indexing="ij")
image = torch.stack([xx, yy, 0.5 * (xx + yy)], -1).to(dev)
kw, ts = make(True)
assert ts.bucket is not None and ts.world == world
# the ranks' pixels are disjoint slices of ONE global sample: global ray ids rank * N_RAND ... (rank + 1) * N_RAND - 1
_, _, pix = P.select_view_rays(H, W, K, c2w, image, N_RAND, 2.0, 6.0, seed=3, step=0, ray_id0=rank * N_RAND, want_pixels=True)
flat = (pix[:, 0].long() * W + pix[:, 1].long()).cpu()
every = [None] * world
dist.all_gather_object(every, flat.tolist())
if rank == 0:
    allpix = [q for part in every for q in part]
    assert len(set(allpix)) == world * N_RAND, "the ranks drew overlapping pixels"
    _, _, pix1 = P.select_view_rays(H, W, K, c2w, image, world * N_RAND, 2.0, 6.0, seed=3, step=0, ray_id0=0, want_pixels=True)
    assert (pix1[:, 0].long() * W + pix1[:, 1].long()).cpu().tolist() == allpix, "shards are not slices of the global sample"
losses = []
for step in range(STEPS):
    loss, _ = ts.step_view(H, W, K, c2w, image, near=2.0, far=6.0, n_rand=N_RAND)
    assert ts.bucket.pending() == 0
    # the merged backward leaves both networks' gradients (and their status tails) back to back: ONE collective per step
    assert ts.bucket.collectives == 1, ts.bucket.collectives
    losses.append(float(loss))
digest = [float(p.detach().double().sum()) for n in ts.nets for p in n.parameters()]
gathered = [None] * world
dist.all_gather_object(gathered, digest)
assert all(gd == gathered[0] for gd in gathered), "replicas diverged"
if rank == 0:
    # the same steps as ONE process over the global batch of world * N_RAND rays (same seed: same pixels, same draws)
    kw1, ts1 = make(False)
    for step in range(STEPS):
        ts1.step_view(H, W, K, c2w, image, near=2.0, far=6.0, n_rand=world * N_RAND)
    worst = max(float((p.detach() - q.detach()).abs().max()) for n, m in zip(ts.nets, ts1.nets)
                for p, q in zip(n.parameters(), m.parameters()))
    print(f"max |param({world} ranks x {N_RAND} rays) - param(1 rank x {world * N_RAND} rays)| =", worst)
    assert worst <= 2e-4, worst          # three Adam steps (lr 5e-4) apart at most through rounding-level gradient differences
    del kw1, ts1
    torch.cuda.empty_cache()
# The range guard is global: SOME ranks' forward leaves the half range (simulated: their fine network's status word is set
# as the clamping kernel would set it) -> the word travels as the tail element of the fine network's gradient buffer, the
# summed tail guards Adam on EVERY rank (no extra collective), all withhold the step, all raise at the next check --
# the ranks whose own words are clear because their optimizer counted a withheld step -- and the step counts are wound back.
before = [p.detach().clone() for n in ts.nets for p in n.parameters()]
steps_before = float(ts.optimizer.state[next(ts.nets[1].parameters())]['step'])
if rank in FLAGGED:
    ts.nets[1].status_word().fill_(1)
ts.step_view(H, W, K, c2w, image, near=2.0, far=6.0, n_rand=N_RAND)
fine_before = before[len(list(ts.nets[0].parameters())):]
assert all(torch.equal(a, p.detach()) for a, p in zip(fine_before, ts.nets[1].parameters())), "a guarded step reached the weights"
assert int(ts.nets[1].status_word().item()) == int(rank in FLAGGED), f"rank {rank}: status words are per rank (the tails carry them)"
tails = ts.bucket.tails()
assert len(tails) == 2 and float(tails[0]) == 0.0 and float(tails[1]) == float(len(FLAGGED)), [float(t) for t in tails]
raised = None
try:
    ts.check_range()
except FloatingPointError as e:
    raised = str(e)
assert raised is not None, f"rank {rank} did not raise"
assert ("another rank" in raised) == (rank not in FLAGGED), raised
assert float(ts.optimizer.state[next(ts.nets[1].parameters())]['step']) == steps_before       # wound back
# (the coarse network's word was clear on every rank: its step went through, identically)
digest = [float(p.detach().double().sum()) for p in ts.nets[0].parameters()]
gathered = [None] * world
dist.all_gather_object(gathered, digest)
assert all(gd == gathered[0] for gd in gathered)
if world == 2:
    # The FALLBACK exchange (ADVICE r04): a batch split over several render_rays / MlpFn calls (chunk < N_rand) leaves
    # `.grad` as a sum of buffers -- no flat buffer, no status tail.  The ranks' status words then go through one MAX
    # all-reduce (dp.GradientBucket._sync_status): rank 1's coarse forward "clamps", BOTH ranks withhold the coarse step,
    # both see the word set, both raise.
    for n in ts.nets:
        n.status_word().zero_()
    ts.args.chunk = N_RAND // 2
    before = [p.detach().clone() for p in ts.nets[0].parameters()]
    if rank == 1:
        ts.nets[0].status_word().fill_(1)
    ts.step_view(H, W, K, c2w, image, near=2.0, far=6.0, n_rand=N_RAND)
    assert ts.bucket.collectives >= 2, ts.bucket.collectives          # the gathered bucket + the status words
    assert all(torch.equal(a, p.detach()) for a, p in zip(before, ts.nets[0].parameters())), "fallback: a guarded step reached the weights"
    assert int(ts.nets[0].status_word().item()) == 1, f"rank {rank}: the fallback writes the ranks' MAX back"
    try:
        ts.check_range()
        raise SystemExit(f"rank {rank} did not raise on the fallback path")
    except FloatingPointError:
        pass
print(f"rank {rank} ok")
dist.destroy_process_group()
"""


def _run_dp_ranks(tmp_path, world, hw, n_rand, timeout):
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "dp_gpu_worker.py"
    script.write_text(_DP_GPU_WORKER)
    port = 29700 + (os.getpid() % 200) + 200 * (world > 2)
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script), root, str(hw), str(n_rand)], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=timeout)[0] for p in procs]
    for rank, (p, out) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, f"rank {rank} failed:\n{out[-3000:]}"
        assert f"rank {rank} ok" in out
    print([l for l in outs[0].splitlines() if l.startswith("max |param")])


def test_data_parallel_training_step_two_ranks_on_one_gpu(P, tmp_path):
    """The multi-GPU step as the driver will launch it, minus the second GPU: two ranks (gloo backend, both on cuda:0)
    each render their shard of a global batch chosen by the counter-based generator, the post-accumulate hooks enqueue
    one in-place all-reduce per network from inside backward, the guarded flat Adam steps.  Replicas stay bit-identical
    over three steps, and the weights land where ONE process stepping the whole global batch lands (same pixels, same
    draws -- world-size invariance).  Also the fallback exchange's status synchronisation."""
    _run_dp_ranks(tmp_path, 2, 64, 128, 600)


def test_baseline_config2_eight_shards_equal_one_global_batch(P, tmp_path):
    """BASELINE configs[2] in its real shape, minus the other seven GPUs: EIGHT ranks x 4096 rays of an 800 x 800 view
    (global batch 32,768; global ray ids up to 32,767 through the counter-based pixel choice and draws; 1 / world = 1 / 8
    inside the Adam kernel; status tails summed over eight ranks), time-sharing one MI355X over gloo, against ONE rank
    stepping the whole 32,768-ray batch: parameters after three steps equal to fp32 summation-order tolerance.  What
    this leaves untested of configs[2] is the RCCL wire itself."""
    _run_dp_ranks(tmp_path, 8, 800, 4096, 1500)


# ----------------------------------------------------------------------------- the depth-supervised step, data parallel
_DP_DEPTH_WORKER = r"""
import os, sys, torch, torch.distributed as dist
from argparse import Namespace
sys.path.insert(0, sys.argv[1])
sys.path.insert(0, os.path.join(sys.argv[1], "tests"))
import plnerf_amd as P
from plnerf_amd import dp, depth as Dp, functional as Fn
from oracle import plnerf_oracle as orc
HW, N_RAND, JOINT, STEPS = int(sys.argv[2]), int(sys.argv[3]), bool(int(sys.argv[4])), 3
rank, world, _ = dp.init_from_env(backend="gloo")          # every rank on cuda:0; gloo moves CUDA tensors through the host
dev = torch.device("cuda:0")
torch.cuda.set_device(0)
# BASELINE configs[4]'s sampling: N_samples 128 / N_importance 64, mode linear, depth loss; the 57 | 3-channel network
ARGS = dict(multires=9, i_embed=0, use_viewdirs=True, multires_views=0, input_ch_cam=0, N_importance=64, N_samples=128,
            netdepth=8, netwidth=256, netdepth_fine=8, netwidth_fine=256, netchunk=65536, lrate=5e-4, perturb=1.0,
            white_bkgd=True, raw_noise_std=0.0, mode="linear", color_mode="midpoint", lindisp=False, no_reload=True,
            space_carving_weight=0.007, warm_start_nerf=0, is_joint=JOINT, norm_p=2, space_carving_threshold=0.0,
            precision="f16x3", bb_center=0.0, bb_scale=1.0)


def make(distributed):
    args = Namespace(**ARGS)
    kw, _, _, grad_vars, opt = Dp.create_nerf(args, device=dev)
    kw["network_fn"].load_state_dict(orc.closed_form_state_dict_depth(0, True))
    kw["network_fine"].load_state_dict(orc.closed_form_state_dict_depth(1, True))
    return Dp.DepthTrainStep(args, kw, opt, grad_vars, distributed=distributed, seed=3)


H = W = HW
K = [[1.4 * HW, 0, W / 2], [0, 1.4 * HW, H / 2], [0, 0, 1]]
c2w = P.rays.pose_spherical(20.0, -30.0, 4.0)[:3, :4]
yy, xx = torch.meshgrid(torch.linspace(0, 1, H), torch.linspace(0, 1, W), indexing="ij")
image = torch.stack([xx, yy, 0.5 * (xx + yy)], -1).to(dev)
hyp_img = (2.0 + 4.0 * torch.rand(3, H, W, generator=torch.Generator().manual_seed(11))).to(dev)      # three depth hypotheses per pixel


def batch_of(step, n, ray_id0):
    cols, target, pix = P.select_view_rays(H, W, K, c2w, image, n, 2.0, 6.0, seed=3, step=step, ray_id0=ray_id0, want_pixels=True)
    return cols, target, hyp_img[:, pix[:, 0].long(), pix[:, 1].long()].unsqueeze(-1)


if JOINT:
    # the hypothesis choice itself, on data where the shards disagree: every rank holds its slice of ONE global pred_hyp;
    # the sharded choice (column sums + one all-reduce) must be the one-process choice over all rays, and the oracle's
    gen = torch.Generator().manual_seed(5)
    R_all, NP = world * 128, 64
    pred = 2.0 + 4.0 * torch.rand(R_all, NP, generator=gen)
    th = 2.0 + 4.0 * torch.rand(3, R_all, 1, generator=gen)
    sl = slice(rank * 128, (rank + 1) * 128)
    mine = Fn.joint_choice(pred[sl].to(dev), th[:, sl].contiguous().to(dev), None, 0.0, None).cpu()
    dist.barrier()
    was = dist.group.WORLD
    whole = (pred[None] - th).abs().mean(1).argmin(0).to(torch.int32)                # model/run_nerf_helpers.py:72-77
    local = (pred[None, sl] - th[:, sl]).abs().mean(1).argmin(0).to(torch.int32)
    assert torch.equal(mine, whole), (rank, int((mine != whole).sum()))
    n_differ = int((local != whole).sum())
    every = [None] * world
    dist.all_gather_object(every, n_differ)
    if rank == 0:
        assert sum(every) > 0, "the shards' own choices all equal the global one: this data does not test the exchange"
        print(f"is_joint: {sum(every)} of {world * NP} per-shard choices differ from the global batch's; the sharded choice equals it on every rank")

ts = make(True)
assert ts.bucket is not None
losses = []
for step in range(STEPS):
    cols, target, target_h = batch_of(step, N_RAND, rank * N_RAND)
    loss, img_loss, sc, _ = ts(cols, target, target_h)
    assert ts.bucket.pending() == 0
    assert ts.bucket.collectives == 1, ts.bucket.collectives      # the merged backward: both networks' gradients + tails, one exchange
    losses.append((float(loss), float(sc)))
digest = [float(p.detach().double().sum()) for n in ts.nets for p in n.parameters()]
gathered = [None] * world
dist.all_gather_object(gathered, (digest, losses))
assert all(gd[0] == gathered[0][0] for gd in gathered), "replicas diverged"
if rank == 0:
    # ONE process over the global batch of world * N_RAND rays: same pixels, same draws, clip_value on the same (averaged) gradient
    ts1 = make(False)
    one = []
    for step in range(STEPS):
        cols, target, target_h = batch_of(step, world * N_RAND, 0)
        loss, img_loss, sc, _ = ts1(cols, target, target_h)
        one.append((float(loss), float(sc)))
    mean_dp = [tuple(sum(gd[1][k][j] for gd in gathered) / world for j in range(2)) for k in range(STEPS)]
    print("loss / space carving, mean over ranks vs one rank:", mean_dp, one)
    for (a, sa), (b, sb) in zip(mean_dp, one):      # (the ranks' losses are means over their shards: their average is the global mean)
        assert abs(a - b) <= 1e-5 * max(1.0, abs(b)) and abs(sa - sb) <= 1e-4 * max(1.0, abs(sb)), (mean_dp, one)
    worst = max(float((p.detach() - q.detach()).abs().max()) for n, m in zip(ts.nets, ts1.nets)
                for p, q in zip(n.parameters(), m.parameters()))
    print(f"depth step: max |param({world} ranks x {N_RAND} rays) - param(1 rank x {world * N_RAND} rays)| =", worst)
    # (Adam moves every weight by ~lr = 5e-4 per step whatever its gradient's size: an entry whose gradient is ~0 may flip sign
    # under another summation order -- 2 lr per step, the bound of the one-GPU merged-vs-autograd comparison)
    assert worst <= 1.1e-3, worst
    del ts1
    torch.cuda.empty_cache()
# the range guard travels with the one exchange here too: a flagged rank withholds the single Adam over both networks everywhere
before = [p.detach().clone() for n in ts.nets for p in n.parameters()]
if rank == world - 1:
    ts.nets[1].status_word().fill_(1)
cols, target, target_h = batch_of(STEPS, N_RAND, rank * N_RAND)
ts(cols, target, target_h)
assert all(torch.equal(a, p.detach()) for a, p in zip(before, (p for n in ts.nets for p in n.parameters()))), "a guarded step reached the weights"
tails = ts.bucket.tails()
assert len(tails) == 2 and float(tails[0]) == 0.0 and float(tails[1]) == 1.0, [float(t) for t in tails]
try:
    ts.check_range()
    raise SystemExit(f"rank {rank} did not raise")
except FloatingPointError as e:
    assert ("another rank" in str(e)) == (rank != world - 1), str(e)
print(f"rank {rank} ok")
dist.destroy_process_group()
"""


def _run_dp_depth_ranks(tmp_path, world, hw, n_rand, joint, timeout):
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "dp_depth_worker.py"
    script.write_text(_DP_DEPTH_WORKER)
    port = 28700 + (os.getpid() % 200) + 200 * (world > 2) + 400 * joint
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script), root, str(hw), str(n_rand), str(int(joint))], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=timeout)[0] for p in procs]
    for rank, (p, out) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, f"rank {rank} failed:\n{out[-3000:]}"
        assert f"rank {rank} ok" in out
    print([l for l in outs[0].splitlines() if l.startswith(("depth step", "is_joint", "loss /"))])


@pytest.mark.parametrize("joint", [False, True])
def test_depth_supervised_step_data_parallel_two_ranks_on_one_gpu(P, tmp_path, joint):
    """DepthTrainStep's data-parallel branch (run_nerf_sample_based_depth.py:1126-1157 on ray shards; the reference runs this
    variant under nn.DataParallel, :564, 585): two gloo ranks on cuda:0, each with its shard of a global batch (disjoint
    slices of one pixel sample, counter-based draws keyed on the global ray id), ONE collective per step for both networks'
    gradients and status tails, 1 / world and clip_value = 0.1 on the averaged gradient inside the one Adam launch.  Replicas
    bit-identical; parameters and losses equal to ONE process stepping the global batch.  joint: the is_joint loss chooses
    its hypothesis per point column from the mean over the WHOLE batch (model/run_nerf_helpers.py:72-77) -- the shards'
    column sums are added before the minimum (functional.joint_choice), checked on data where the shards alone would choose
    otherwise."""
    _run_dp_depth_ranks(tmp_path, 2, 64, 128, joint, 600)


def test_baseline_config4_eight_shards_equal_one_global_batch(P, tmp_path):
    """BASELINE configs[4] in its real shape, minus the other seven GPUs: the depth-supervised step at N_samples 128 /
    N_importance 64, EIGHT ranks x 4096 rays of an 800 x 800 view time-sharing one MI355X over gloo, against ONE rank stepping
    the 32,768-ray batch (clip on the averaged gradient, tails summed over eight ranks, one exchange per step).  What this leaves
    untested of configs[4] is the RCCL wire."""
    _run_dp_depth_ranks(tmp_path, 8, 800, 4096, False, 1500)


@pytest.mark.parametrize("precision,n_rows", [("f16x3", 131072), ("f16x3", 262144), ("f16x3", 1000), ("fp32", 8200),
                                               ("fp32", 57400)])
def test_weight_gradients_do_not_depend_on_what_the_workspace_held(P, precision, n_rows):
    """Regression test of round 4's find: plnerf_mlp_bwd's split-K reduction must add only partials that a workgroup
    wrote.  Rounding a row range's length up to whole stages used to leave the last ranges without a row (131,072 rows
    over 85 ranges of 1,600: ranges 82-84; 262,144 -- the benchmark's coarse pass -- range 84), whose never-written
    partials were summed into four gradient tensors.  Through the C ABI with a workspace the caller poisoned with NaN
    against one it zeroed: the 24 gradients must be finite and bit-identical."""
    import ctypes
    from plnerf_amd import _lib as L
    net = make_net(P, orc.closed_form_state_dict(0, False), precision)
    prec = L.PRECISION[precision]
    gen = torch.Generator().manual_seed(4)
    spr = 8
    n_rays = (n_rows + spr - 1) // spr
    n_rows = n_rays * spr
    pts = g((torch.rand(n_rows, 3, generator=gen) * 2 - 1) * 2.0)
    vd = g(torch.nn.functional.normalize(torch.randn(n_rays, 3, generator=gen), dim=-1))
    g_raw = g(torch.randn(n_rows, 4, generator=gen))
    packed = net.packed_weights()
    raw = torch.empty(n_rows, 4, device=dev())
    saved = torch.empty(L.lib().plnerf_mlp_saved_bytes(n_rows, prec) // 4, device=dev())
    L.check(L.lib().plnerf_mlp_fwd(L.dptr(packed), prec, L.dptr(pts), L.dptr(vd), None, 63, 27, n_rows, spr, 1.0, 0.0,
                                   L.dptr(raw), L.dptr(saved), L.FWD_KERNEL, L.stream()), "plnerf_mlp_fwd")
    layout = L.lib().plnerf_mlp_saved_layout(prec, 0, L.FWD_KERNEL)
    shapes = [p.shape for p in net.parameters()]
    results = []
    for fill in (float("nan"), 0.0):
        ws = torch.full((L.lib().plnerf_mlp_bwd_workspace_bytes(n_rows, prec) // 4,), fill, device=dev())
        grads = [torch.full(tuple(s), float("nan"), device=dev()) for s in shapes]
        L.check(L.lib().plnerf_mlp_bwd(L.dptr(packed), prec, L.dptr(g_raw), None, 0, 63, 27, n_rows, L.dptr(saved), layout, None, 0.0,
                                       L.dptr(ws), L.ptr_table(grads, "grads"), None, L.stream()), "plnerf_mlp_bwd")
        torch.cuda.synchronize()
        results.append(grads)
    for (name, _), a, b in zip(net.named_parameters(), *results):
        assert torch.isfinite(a).all(), f"{name}: non-finite gradient entries out of a NaN-poisoned workspace"
        assert torch.equal(a, b), f"{name}: the gradient depends on the workspace's previous contents"


# ----------------------------------------------------------------------------- both networks' backward in one launch sequence
def test_merged_backward_equals_two_backwards(P):
    """plnerf_mlp_bwd_multi (ABI 500): the coarse and the fine network's backward as ONE launch sequence -- one gradient-
    chain grid, one launch of each weight-gradient kernel, the round of workgroups dealt out over the two networks by
    rows -- against the two separate plnerf_mlp_bwd calls, through the C ABI on poisoned workspaces, at the benchmark's
    sizes (262,144 + 786,432 rows), at ragged ones, and with the roles swapped.  The gradient chain's arithmetic does not
    depend on the grid (dz planes bit-identical => bias gradients and head gradients bit-identical); the 256-wide weight
    gradients sum their split-K partials over other row ranges (21 + 7 instead of 28 + 28), so they agree to fp32
    summation order.  Also: max |g_raw| handed in as an array of partial maxima (g_absmax / n_absmax: what plnerf_quad_bwd
    leaves per workgroup) instead of found by the call's own pass = the same bits."""
    import ctypes
    from plnerf_amd import _lib as L
    prec = L.PRECISION["f16x3"]
    nets = [make_net(P, orc.closed_form_state_dict(k, False), "f16x3") for k in (0, 1)]
    gen = torch.Generator().manual_seed(6)

    def forward(net, n_rays, spr):
        n_rows = n_rays * spr
        pts = g((torch.rand(n_rows, 3, generator=gen) * 2 - 1) * 2.0)
        vd = g(torch.nn.functional.normalize(torch.randn(n_rays, 3, generator=gen), dim=-1))
        g_raw = g(torch.randn(n_rows, 4, generator=gen) * 1e-3)
        packed = net.packed_weights()
        raw = torch.empty(n_rows, 4, device=dev())
        saved = torch.empty(L.lib().plnerf_mlp_saved_bytes(n_rows, prec) // 4, device=dev())
        L.check(L.lib().plnerf_mlp_fwd(L.dptr(packed), prec, L.dptr(pts), L.dptr(vd), None, 63, 27, n_rows, spr, 1.0, 0.0,
                                       L.dptr(raw), L.dptr(saved), L.FWD_KERNEL, L.stream()), "plnerf_mlp_fwd")
        return dict(packed=packed, g_raw=g_raw, n_rows=n_rows, saved=saved)
    layout = L.lib().plnerf_mlp_saved_layout(prec, 0, L.FWD_KERNEL)
    shapes = [p.shape for p in nets[0].parameters()]
    names = [n for n, _ in nets[0].named_parameters()]
    vp = lambda items: (ctypes.c_void_p * len(items))(*[None if x is None else x.value for x in items])

    def fresh(job):
        ws = torch.full((L.lib().plnerf_mlp_bwd_workspace_bytes(job["n_rows"], prec) // 4,), float("nan"), device=dev())
        return ws, [torch.full(tuple(s), float("nan"), device=dev()) for s in shapes]

    for sizes in ((4096 * 64, 4096 * 192), (4096 * 192, 4096 * 64), (1000 * 8, 37 * 8), (300 * 8, 90000 * 8)):
        jobs = [forward(net, n // 8, 8) for net, n in zip(nets, sizes)]
        single = []
        for job in jobs:
            ws, grads = fresh(job)
            L.check(L.lib().plnerf_mlp_bwd(L.dptr(job["packed"]), prec, L.dptr(job["g_raw"]), None, 0, 63, 27, job["n_rows"],
                                           L.dptr(job["saved"]), layout, None, 0.0, L.dptr(ws), L.ptr_table(grads, "grads"), None,
                                           L.stream()), "plnerf_mlp_bwd")
            single.append((ws, grads))
        for with_absmax in (False, True):
            both = [fresh(job) for job in jobs]
            am = [None, None]
            if with_absmax:      # per-group maxima as fp32 bits, as plnerf_quad_bwd's absmax_out leaves them (here: per 8 rows, ragged)
                am = [torch.nn.functional.pad(job["g_raw"].abs().reshape(-1), (0, (-job["g_raw"].numel()) % 32)).reshape(-1, 32)
                      .max(-1).values.contiguous().view(torch.int32) for job in jobs]
            tails = torch.full((2, 4), -1.0, device=dev())
            L.check(L.lib().plnerf_mlp_bwd_multi(
                2, vp([L.dptr(j["packed"]) for j in jobs]), prec, vp([L.dptr(j["g_raw"]) for j in jobs]),
                vp([L.dptr(a, "g_absmax", torch.int32) for a in am]), (ctypes.c_int * 2)(*[0 if a is None else a.numel() for a in am]),
                63, 27, (ctypes.c_int * 2)(*[j["n_rows"] for j in jobs]),
                vp([L.dptr(j["saved"]) for j in jobs]), (ctypes.c_int * 2)(layout, layout), None, 0.0,
                vp([L.dptr(ws) for ws, _ in both]), L.ptr_table([t for _, gr in both for t in gr], "grads"),
                vp([ctypes.c_void_p(tails[k].data_ptr()) for k in range(2)]), L.stream()), "plnerf_mlp_bwd_multi")
            torch.cuda.synchronize()
            assert tails[:, 0].tolist() == [0.0, 0.0]
            worst = 0.0
            for k, job in enumerate(jobs):
                # the dz planes (the workspace's first section): the gradient chain is the same arithmetic in either grid
                n_dz = 4352 * ((job["n_rows"] + 191) // 192 * 192) // 4
                assert torch.equal(single[k][0][:n_dz].view(torch.int32), both[k][0][:n_dz].view(torch.int32)), (sizes, k, "dz planes")
                for name, a, b in zip(names, single[k][1], both[k][1]):
                    assert torch.isfinite(b).all(), (sizes, k, name)
                    scale = float(a.abs().max()) + 1e-30
                    err = float((a - b).abs().max()) / scale
                    worst = max(worst, err)
                    # (column sums: per-range partials, other ranges.  alpha_linear.weight is a split-K MFMA product like the 256-wide
                    # weights since round 6 -- g_sigma^T h7 on the view job's idle waves -- and takes their bound)
                    if name.endswith("bias") or name == "alpha_linear.bias" or name.startswith("rgb_linear"):
                        assert torch.equal(a, b) or err <= 2e-6, (sizes, k, name, err)
                    assert err <= 2e-5, (sizes, k, name, err)
            print(f"merged vs separate backward, rows {sizes}, g_absmax handed in {with_absmax}: worst gradient difference "
                  f"{worst:.2e} of max|g| (fp32 summation order of the split-K partials)")


def test_multi_backward_in_exact_fp32_is_the_jobs_in_turn(P):
    """plnerf_mlp_bwd_multi in PLNERF_PREC_FP32 runs its jobs one after the other on the exact-fp32 kernels: bit-identical to
    separate plnerf_mlp_bwd calls (and the argument checks: more than PLNERF_MAX_BWD_JOBS jobs, a null entry)."""
    import ctypes
    from plnerf_amd import _lib as L
    prec = L.PRECISION["fp32"]
    nets = [make_net(P, orc.closed_form_state_dict(k, False), "fp32") for k in (0, 1)]
    gen = torch.Generator().manual_seed(8)
    vp = lambda items: (ctypes.c_void_p * len(items))(*[None if x is None else x.value for x in items])
    jobs = []
    for net, n_rays in zip(nets, (700, 231)):
        n_rows = n_rays * 8
        pts = g((torch.rand(n_rows, 3, generator=gen) * 2 - 1) * 2.0)
        vd = g(torch.nn.functional.normalize(torch.randn(n_rays, 3, generator=gen), dim=-1))
        packed = net.packed_weights()
        raw = torch.empty(n_rows, 4, device=dev())
        saved = torch.empty(L.lib().plnerf_mlp_saved_bytes(n_rows, prec) // 4, device=dev())
        L.check(L.lib().plnerf_mlp_fwd(L.dptr(packed), prec, L.dptr(pts), L.dptr(vd), None, 63, 27, n_rows, 8, 1.0, 0.0, L.dptr(raw),
                                       L.dptr(saved), L.FWD_KERNEL, L.stream()), "plnerf_mlp_fwd")
        jobs.append(dict(packed=packed, g_raw=g(torch.randn(n_rows, 4, generator=gen)), n_rows=n_rows, saved=saved))
    shapes = [p.shape for p in nets[0].parameters()]
    fresh = lambda job: (torch.zeros(L.lib().plnerf_mlp_bwd_workspace_bytes(job["n_rows"], prec) // 4, device=dev()),
                         [torch.full(tuple(s), float("nan"), device=dev()) for s in shapes])
    single = []
    for job in jobs:
        ws, grads = fresh(job)
        L.check(L.lib().plnerf_mlp_bwd(L.dptr(job["packed"]), prec, L.dptr(job["g_raw"]), None, 0, 63, 27, job["n_rows"],
                                       L.dptr(job["saved"]), 0, None, 0.0, L.dptr(ws), L.ptr_table(grads, "grads"), None, L.stream()),
                "plnerf_mlp_bwd")
        single.append(grads)
    both = [fresh(job) for job in jobs]
    call = lambda n: L.lib().plnerf_mlp_bwd_multi(
        n, vp([L.dptr(j["packed"]) for j in jobs]), prec, vp([L.dptr(j["g_raw"]) for j in jobs]), None, None, 63, 27,
        (ctypes.c_int * 2)(*[j["n_rows"] for j in jobs]), vp([L.dptr(j["saved"]) for j in jobs]), (ctypes.c_int * 2)(0, 0), None, 0.0,
        vp([L.dptr(ws) for ws, _ in both]), L.ptr_table([t for _, gr in both for t in gr], "grads"), None, L.stream())
    L.check(call(2), "plnerf_mlp_bwd_multi")
    torch.cuda.synchronize()
    for k in range(2):
        for a, b in zip(single[k], both[k][1]):
            assert torch.equal(a, b)
    assert call(3) == -1 and call(0) == -1      # PLNERF_EINVAL


def test_training_step_with_merged_backward_equals_autograd_order(P):
    """train.TrainStep's merged backward (autograd down to d loss / d raw of both networks with max |g_raw| as
    plnerf_quad_bwd's by-product, then plnerf_mlp_bwd_multi, `.grad` assigned directly) against the same steps through
    torch.autograd.backward (PLNERF_MERGED_BWD = 0's route): same losses to fp32 summation order, same weights after five
    steps to an Adam step's rounding, every backward job took its maximum from the by-product."""
    from plnerf_amd import functional as Fn
    H = W = 200
    K = [[280.0, 0, W / 2], [0, 280.0, H / 2], [0, 0, 1]]
    image = g(torch.rand(H, W, 3, generator=torch.Generator().manual_seed(3)))
    poses = [P.rays.pose_spherical(-180.0 + 90.0 * i, -30.0, 4.0)[:3, :4] for i in range(4)]

    def run(merged):
        args, kw, opt, opt_c = _nets(P)
        ts = P.TrainStep(args, kw, opt, opt_c, distributed=False, seed=11)
        ts.merged_backward = merged
        hits0 = Fn.ABSMAX_HITS
        losses = []
        for step in range(5):
            loss, _ = ts.step_view(H, W, K, poses[step % 4], image, near=2.0, far=6.0, n_rand=4096)
            losses.append(float(loss))
        return losses, [p.detach().clone() for n in ts.nets for p in n.parameters()], Fn.ABSMAX_HITS - hits0
    l1, p1, hits1 = run(True)
    l0, p0, hits0 = run(False)
    assert hits1 == 10 and hits0 == 0, (hits1, hits0)
    for a, b in zip(l1, l0):
        assert abs(a - b) <= 2e-6 * max(1.0, abs(b)), (l1, l0)
    worst = max(float((a - b).abs().max()) for a, b in zip(p1, p0))
    print(f"merged vs autograd-order backward, 5 steps x 4096 rays: losses {l1[-1]:.7f} / {l0[-1]:.7f}, max parameter difference {worst:.2e}")
    assert worst <= 2e-4, worst
