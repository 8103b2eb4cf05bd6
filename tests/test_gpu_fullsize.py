"""Oracle parity at BASELINE.json's FULL per-GPU size (missing #3): 4096 rays x (64 + 192) samples -- the
configuration bench.py times (configs[1]) -- and the config file's own 128 + 64 sampling, in the benchmarked arithmetic
(f16x3) and in exact fp32, against the CPU oracle on identical `pytest=True` draws (the reference's np.random.seed(0)
draws, run_plnerf.py:700-703, run_nerf_helpers.py:384-392).  The small fixtures (32-96 rays) cannot see a tile-boundary,
grid-size or padded-plane bug of a 6,144-workgroup launch; these can.

Stated tolerances:
  * coarse maps (continuous in the network output): 1e-5 abs+rel on EVERY ray -- the contract;
  * final maps pass through the sampler, which is discontinuous (a fine sample hops a cdf bin when the
    coarse network's output moves by an ulp): the NUMBER of rays beyond 1e-5 is asserted and printed per map --
    0 on rgb; acc, depth and z_std may have a handful of hopping rays out of 4096 (bounds below; the exact-fp32 kernels
    show the same handful -- measured: fp32 64+128 depth 5 rays, 128+64 depth 1 / z_std 2; f16x3 64+128 acc 1 / depth 9, 128+64 depth 1 / z_std 2);
  * one full-size training step: loss to 1e-5; every parameter tensor's gradient within the bound of the small-fixture
    tests (fp32: 2e-4 coarse / 2e-3 fine of max|g|; f16x3: 6e-3 / 3e-3, the half-plane backward).

Round 4 adds the two other single-GPU workloads of BASELINE.json at full size against the oracle: configs[3] (LLFF:
4096 NDC rays in [near 0, far 1], raw_noise_std = 1 with the reference's `pytest` noise draw, no white background) and
configs[4]'s render (the depth-supervised variant: 57|3-channel network with the pi-scaled encoding, softplus density,
128 + 64 samples, pred_hyp) -- same structure: coarse maps at 1e-5 on every ray, final maps by count.

Host cost: four oracle renders (about 5 s each on 16 threads) and one oracle training step (about 20 s).
"""
import os

import pytest
import torch

from oracle import plnerf_oracle as orc
from test_gpu_parity import assert_close, g, make_net, maxdiff

pytestmark = pytest.mark.gpu
R_FULL = 4096
SAMPLINGS = [(64, 128), (128, 64)]
# rays (of 4096) allowed beyond 1e-5 per final map: rgb admits none; a hopping fine sample moves acc / depth / z_std of
# its ray (the kernels are deterministic and the draws fixed, so the counts are reproducible box to box)
# Per precision and sampling: the measured count + 2 (a regression that doubles the hopping rays must fail).
# (PLNERF_FWD_KERNEL=pp -- the suite's pass over the ping-pong forward kernels: another fp32 summation order, so OTHER
# samples sit within rounding of a bin edge; the counts above were measured on the default kernels)
OTHER_KERNEL_SLACK = 2 if os.environ.get("PLNERF_FWD_KERNEL", "") == "pp" else 0
MAX_RAYS_BEYOND = {
    ("f16x3", 64, 128): {"rgb_map": 0, "acc_map": 3, "depth_map": 11, "z_std": 2},
    ("f16x3", 128, 64): {"rgb_map": 0, "acc_map": 2, "depth_map": 3, "z_std": 4},
    ("fp32", 64, 128): {"rgb_map": 0, "acc_map": 2, "depth_map": 7, "z_std": 2},
    ("fp32", 128, 64): {"rgb_map": 0, "acc_map": 2, "depth_map": 3, "z_std": 4},
}


@pytest.fixture(scope="module")
def P():
    import plnerf_amd
    return plnerf_amd


@pytest.fixture(scope="module")
def oracle_renders():
    """The oracle's render of the full-size batch, once per sampling (shared by the precision legs)."""
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    cache = {}

    def get(ns, ni):
        if (ns, ni) not in cache:
            batch, _ = orc.synthetic_blender_rays(R_FULL, seed=11)
            sd_c, sd_f = orc.closed_form_state_dict(0, True), orc.closed_form_state_dict(1, True)
            with torch.no_grad():
                ref = orc.render_rays(batch, sd_c, sd_f, ns, "linear", "midpoint", retraw=True, perturb=1.0,
                                      N_importance=ni, white_bkgd=True, pytest=True)
            cache[(ns, ni)] = (batch, sd_c, sd_f, ref)
        return cache[(ns, ni)]
    return get


@pytest.mark.parametrize("precision", ["f16x3", "fp32"])
@pytest.mark.parametrize("sampling", SAMPLINGS, ids=lambda s: f"{s[0]}+{s[1]}")
def test_render_rays_at_baseline_size_vs_oracle(P, oracle_renders, sampling, precision):
    ns, ni = sampling
    batch, sd_c, sd_f, ref = oracle_renders(ns, ni)
    emb_fn, _ = P.get_embedder(10, 0)
    embd_fn, _ = P.get_embedder(4, 0)
    qfn = lambda inputs, viewdirs, fn: P.run_network(inputs, viewdirs, fn, emb_fn, embd_fn)
    net_c, net_f = make_net(P, sd_c, precision), make_net(P, sd_f, precision)
    with torch.no_grad():
        got = P.render_rays(g(batch), net_c, qfn, ns, "linear", "midpoint", retraw=True, perturb=1.0, N_importance=ni,
                            network_fine=net_f, white_bkgd=True, pytest=True)
    torch.cuda.synchronize()
    if precision != "fp32":
        assert net_c.range_status() == 0 and net_f.range_status() == 0
    # coarse pass: every ray inside the contract
    for k in ("rgb0", "acc0", "depth0", "disp0"):
        assert_close(got[k], ref[k], what=f"{precision} {ns}+{ni} {k}")
    report, over = [], []
    for k, allowed in MAX_RAYS_BEYOND[(precision, ns, ni)].items():
        d = (got[k].cpu() - ref[k]).abs()
        lim = 1e-5 * (1.0 + ref[k].abs())
        bad = d > lim
        n_bad = int((bad.any(-1) if bad.dim() > 1 else bad).sum())
        report.append(f"{k} max {float(d.max()):.2e} beyond {n_bad}")
        if n_bad > allowed + OTHER_KERNEL_SLACK:
            over.append(f"{k}: {n_bad} of {R_FULL} rays beyond 1e-5 (allowed {allowed})")
    print(f"{precision} {ns}+{ni} x {R_FULL} rays: coarse rgb0 {maxdiff(got['rgb0'], ref['rgb0']):.2e}, depth0 "
          f"{maxdiff(got['depth0'], ref['depth0']):.2e}; final " + ", ".join(report))
    assert not over, f"{precision} {ns}+{ni}: " + "; ".join(over)


@pytest.mark.parametrize("precision", ["f16x3", "fp32"])
def test_train_step_at_baseline_size_vs_oracle(P, precision):
    """One optimisation step of configs[1] at full size: loss and all 48 gradient tensors against the oracle's autograd
    on identical rays, targets and draws."""
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    ns, ni = 64, 128
    batch, target = orc.synthetic_blender_rays(R_FULL, seed=12)
    sd_c, sd_f = orc.closed_form_state_dict(0, False), orc.closed_form_state_dict(1, False)
    kw = dict(N_samples=ns, N_importance=ni, mode="linear", color_mode="midpoint", perturb=1.0, white_bkgd=True,
              raw_noise_std=0.0, pytest=True)
    ref_loss, g_c, g_f = orc.train_step({k: v.clone() for k, v in sd_c.items()}, {k: v.clone() for k, v in sd_f.items()},
                                        batch, target, kw)
    emb_fn, _ = P.get_embedder(10, 0)
    embd_fn, _ = P.get_embedder(4, 0)
    qfn = lambda inputs, viewdirs, fn: P.run_network(inputs, viewdirs, fn, emb_fn, embd_fn)
    net_c, net_f = make_net(P, sd_c, precision), make_net(P, sd_f, precision)
    ret = P.render_rays(g(batch), net_c, qfn, retraw=True, network_fine=net_f, **kw)
    loss = P.img2mse(ret["rgb_map"], g(target)) + P.img2mse(ret["rgb0"], g(target))
    loss.backward()
    torch.cuda.synchronize()
    assert abs(float(loss.detach()) - float(ref_loss)) <= 1e-5, (float(loss.detach()), float(ref_loss))
    tol = {"fp32": {"coarse": 2e-4, "fine": 2e-3}, "f16x3": {"coarse": 6e-3, "fine": 3e-3}}[precision]
    worst = {"coarse": 0.0, "fine": 0.0}
    for net, grads, tag in ((net_c, g_c, "coarse"), (net_f, g_f, "fine")):
        for name, prm in net.named_parameters():
            refg = grads[name]
            scale = max(float(refg.abs().max()), 1e-9)
            err = float((prm.grad.cpu() - refg).abs().max())
            worst[tag] = max(worst[tag], err / scale)
            assert err <= tol[tag] * scale + 1e-9, f"{precision} {tag} {name}: grad err {err:.3e} of max|g| {scale:.3e}"
            assert abs(float(prm.grad.norm()) - float(refg.norm())) <= 2e-3 * float(refg.norm()) + 1e-9, (tag, name)
    print(f"{precision} full-size step: loss {float(loss.detach()):.7f} (oracle {float(ref_loss):.7f}); worst grad err / "
          f"max|g|: coarse {worst['coarse']:.2e}, fine {worst['fine']:.2e}")


def _count_beyond(got, ref, tol=1e-5):
    d = (got.cpu() - ref).abs()
    bad = d > tol * (1.0 + ref.abs())
    return int((bad.reshape(bad.shape[0], -1).any(-1)).sum()), float(d.max())


# configs[3]: rays (of 4096) allowed beyond 1e-5 per final map = measured + 2, per precision
# (measured: f16x3 acc 7 / depth 10; fp32 acc 2 / depth 4 / z_std 1 -- the density noise makes more samples hop)
LLFF_MAX_BEYOND = {"f16x3": {"rgb_map": 0, "acc_map": 9, "depth_map": 12, "z_std": 2},
                   "fp32": {"rgb_map": 0, "acc_map": 4, "depth_map": 6, "z_std": 3}}


@pytest.mark.parametrize("precision", ["f16x3", "fp32"])
def test_llff_ndc_render_at_baseline_size_vs_oracle(P, precision):
    """BASELINE configs[3] per GPU: 4096 rays of a 378 x 504 forward-facing view warped to NDC (run_nerf_helpers.py:
    184-201; near 0, far 1 -- run_plnerf.py:1008-1009), density noise raw_noise_std = 1 (the reference's `pytest` draw:
    uniform, run_plnerf.py:573-576), no white background, 64 + 128 samples: the NDC ray range and the noise path through a
    6,144-workgroup launch, against the oracle."""
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    H, W, f = 378, 504, 407.0
    K = [[f, 0, W / 2], [0, f, H / 2], [0, 0, 1]]
    c2w = torch.eye(4)[:3, :4].clone()
    c2w[:, 3] = torch.tensor([0.05, -0.02, 0.1])
    o, d = orc.get_rays(H, W, K, c2w)
    pix = torch.randperm(H * W, generator=torch.Generator().manual_seed(31))[:R_FULL]
    o, d = o.reshape(-1, 3)[pix], d.reshape(-1, 3)[pix]
    vd = d / torch.norm(d, dim=-1, keepdim=True)                      # of the ORIGINAL directions (run_plnerf.py:146-150)
    o_ndc, d_ndc = orc.ndc_rays(H, W, f, 1.0, o, d)
    batch = torch.cat([o_ndc, d_ndc, torch.zeros(R_FULL, 1), torch.ones(R_FULL, 1), vd], -1).float()
    sd_c, sd_f = orc.closed_form_state_dict(0, True), orc.closed_form_state_dict(1, True)
    kw = dict(retraw=True, perturb=1.0, N_importance=128, white_bkgd=False, raw_noise_std=1.0, pytest=True)
    with torch.no_grad():
        ref = orc.render_rays(batch, sd_c, sd_f, 64, "linear", "midpoint", **kw)
    emb_fn, _ = P.get_embedder(10, 0)
    embd_fn, _ = P.get_embedder(4, 0)
    qfn = lambda inputs, viewdirs, fn: P.run_network(inputs, viewdirs, fn, emb_fn, embd_fn)
    net_c, net_f = make_net(P, sd_c, precision), make_net(P, sd_f, precision)
    with torch.no_grad():
        got = P.render_rays(g(batch), net_c, qfn, 64, "linear", "midpoint", network_fine=net_f, **kw)
    torch.cuda.synchronize()
    if precision != "fp32":
        assert net_c.range_status() == 0 and net_f.range_status() == 0
    for k in ("rgb0", "acc0", "depth0", "disp0"):
        assert_close(got[k], ref[k], what=f"{precision} llff_ndc {k}")
    report, over = [], []
    for k, allowed in LLFF_MAX_BEYOND[precision].items():
        n_bad, worst = _count_beyond(got[k], ref[k])
        report.append(f"{k} max {worst:.2e} beyond {n_bad}")
        if n_bad > allowed + OTHER_KERNEL_SLACK:
            over.append(f"{k}: {n_bad} of {R_FULL} rays beyond 1e-5 (allowed {allowed})")
    print(f"{precision} llff_ndc x {R_FULL} rays: coarse rgb0 {maxdiff(got['rgb0'], ref['rgb0']):.2e}, depth0 "
          f"{maxdiff(got['depth0'], ref['depth0']):.2e}; final " + ", ".join(report))
    assert not over, f"{precision} llff_ndc: " + "; ".join(over)


# configs[4]: counts beyond 1e-5 (pred_hyp: beyond 2e-4 -- the sampler's closed form) = measured + 2
# (measured: f16x3 rgb 1 / acc 4 / depth 43 / z_std 46 / pred_hyp 25; fp32 1 / 5 / 53 / 47 / 24: 1 % of the rays -- this network's softplus densities
# are everywhere positive, so far more cdf bins carry weight a sample can hop between than under the NVS networks' relu)
DEPTH_MAX_BEYOND = {"f16x3": {"rgb_map": 3, "acc_map": 6, "depth_map": 45, "z_std": 48, "pred_hyp": 27},
                    "fp32": {"rgb_map": 3, "acc_map": 7, "depth_map": 55, "z_std": 49, "pred_hyp": 26}}


@pytest.mark.parametrize("precision", ["f16x3", "fp32"])
def test_depth_variant_render_at_baseline_size_vs_oracle(P, golden, precision):
    """BASELINE configs[4] per GPU, the render: 4096 rays x (128 + 64) samples through the depth-supervised variant's
    render_rays (run_nerf_sample_based_depth.py:792-958: 57|3-channel network, encoding of x pi 2^k, softplus(beta 10)
    density, pred_hyp from the final weights) -- the `input_scale = pi` route of the fused kernel at 786,432 rows."""
    from test_gpu_modes import _depth_setup
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    gd = golden("g8b_depth_variant_128_64")
    Dp, kw, _, _ = _depth_setup(gd, precision)
    batch, _ = orc.synthetic_blender_rays(R_FULL, seed=13)
    sd_c, sd_f = orc.closed_form_state_dict_depth(0, True), orc.closed_form_state_dict_depth(1, True)
    with torch.no_grad():
        ref = orc.render_rays_depth(batch, sd_c, sd_f, 128, "linear", "midpoint", perturb=1.0, N_importance=64,
                                    white_bkgd=True, pytest=True)
        got = Dp.render_rays(g(batch), retraw=True, pytest=True, **kw)
    torch.cuda.synchronize()
    assert torch.equal(got["u"].cpu(), ref["u"])
    if precision != "fp32":
        assert kw["network_fn"].range_status() == 0 and kw["network_fine"].range_status() == 0
    for k in ("rgb0", "acc0", "depth0", "disp0", "z_vals0", "weights0"):
        assert_close(got[k], ref[k], what=f"{precision} depth_128_64 {k}")
    report, over = [], []
    for k, allowed in DEPTH_MAX_BEYOND[precision].items():
        n_bad, worst = _count_beyond(got[k], ref[k], 2e-4 if k == "pred_hyp" else 1e-5)
        report.append(f"{k} max {worst:.2e} beyond {n_bad}")
        if n_bad > allowed + OTHER_KERNEL_SLACK:
            over.append(f"{k}: {n_bad} of {R_FULL} rays beyond the bound (allowed {allowed})")
    print(f"{precision} depth_128_64 x {R_FULL} rays: coarse rgb0 {maxdiff(got['rgb0'], ref['rgb0']):.2e}, depth0 "
          f"{maxdiff(got['depth0'], ref['depth0']):.2e}; final " + ", ".join(report))
    assert not over, f"{precision} depth_128_64: " + "; ".join(over)
