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

Host cost: two oracle renders (about 5 s each on 16 threads) and one oracle training step (about 20 s).
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
MAX_RAYS_BEYOND = {"rgb_map": 0, "acc_map": 2, "depth_map": 16, "z_std": 8}


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
    for k, allowed in MAX_RAYS_BEYOND.items():
        d = (got[k].cpu() - ref[k]).abs()
        lim = 1e-5 * (1.0 + ref[k].abs())
        bad = d > lim
        n_bad = int((bad.any(-1) if bad.dim() > 1 else bad).sum())
        report.append(f"{k} max {float(d.max()):.2e} beyond {n_bad}")
        if n_bad > allowed:
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
