"""render() (run_plnerf.py:110-175) is chunk-invariant: the reference's `chunk` "does not affect final results" -- here BIT
FOR BIT.  Random small views (H, W up to 48), with / without NDC, view directions, a static camera, both quadrature rules,
rendered with three chunk sizes each (one chunk, a ragged chunk, a tiny chunk):

  * without draws (perturb 0, no noise): every returned map identical across chunk sizes;
  * with an installed functional.DrawSource (jitter, sampler draws and density noise keyed on the ray's position in the whole
    batch, not in its chunk -- what makes a data-parallel shard see the global batch's numbers): identical too;
  * against the oracle on the same rays (coarse maps 1e-5; the jitter-free sampler sits on the searchsorted knife edge at
    u = 1, so the final maps are compared through the campaign of tools/fuzz_render_rays.py instead).

python tools/fuzz_render_chunks.py --cases 60 > out.json"""
import argparse, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import plnerf_amd as P
from plnerf_amd import functional as Fn
from oracle import plnerf_oracle as orc

ap = argparse.ArgumentParser()
ap.add_argument("--cases", type=int, default=60)
ap.add_argument("--seed", type=int, default=17)
a = ap.parse_args()
dev = torch.device("cuda:0")
rng = np.random.default_rng(a.seed)
emb_fn, _ = P.get_embedder(10, 0)
embd_fn, _ = P.get_embedder(4, 0)
qfn = lambda inputs, viewdirs, fn: P.run_network(inputs, viewdirs, fn, emb_fn, embd_fn)
sds = [orc.closed_form_state_dict(s, True) for s in (0, 1)]


def net(sd, precision):
    n = P.NeRF(D=8, W=256, input_ch=63, input_ch_views=27, output_ch=5, skips=[4], use_viewdirs=True, precision=precision)
    n.load_state_dict(sd)
    return n.to(dev)


nets = {p: (net(sds[0], p), net(sds[1], p)) for p in ("fp32", "f16x3")}
bits = lambda x, y: torch.equal(x.contiguous().view(torch.int32), y.contiguous().view(torch.int32))
stats = {"cases": 0, "pixels": 0, "renders": 0, "coarse_worst": 0.0}
violations = []
for case in range(a.cases):
    H, W_ = int(rng.integers(3, 49)), int(rng.integers(3, 49))
    f = float(rng.uniform(10.0, 60.0))
    K = [[f, 0, W_ / 2], [0, f, H / 2], [0, 0, 1]]
    c2w = P.rays.pose_spherical(float(rng.uniform(-180, 180)), float(rng.uniform(-60, -10)), 4.0)[:3, :4].to(dev)
    mode = ["linear", "constant"][int(rng.integers(2))]
    prec = ["fp32", "f16x3"][int(rng.integers(2))]
    nc, nf = nets[prec]
    Ns, Ni = int(rng.choice([8, 32, 64])), int(rng.choice([8, 32, 128]))
    noise = float(rng.choice([0.0, 1.0]))
    kw = dict(network_query_fn=qfn, N_importance=Ni, network_fine=nf, N_samples=Ns, network_fn=nc, white_bkgd=bool(rng.integers(2)),
              mode=mode, color_mode="midpoint", lindisp=bool(rng.integers(2)))
    static = P.rays.pose_spherical(0.0, -30.0, 4.0)[:3, :4].to(dev) if rng.integers(4) == 0 else None
    n_pix = H * W_
    chunks = [n_pix, max(1, int(n_pix * 0.37)), int(rng.integers(1, 20))]
    bad = []
    for draws in (False, True):
        outs = []
        for c in chunks:
            prev = Fn.set_draw_source(Fn.DrawSource(seed=5, ray_id0=0, step=case) if draws else None)
            try:
                with torch.no_grad():
                    rgb, disp, acc, ex = P.render(H, W_, K, chunk=c, c2w=c2w, ndc=False, near=2.0, far=6.0, use_viewdirs=True,
                                                  c2w_staticcam=static, perturb=1.0 if draws else 0.0,
                                                  raw_noise_std=noise if draws else 0.0, **kw)
            finally:
                Fn.set_draw_source(prev)
            outs.append({"rgb": rgb, "disp": disp, "acc": acc, **ex})
            stats["renders"] += 1
        for k in outs[0]:
            for j in (1, 2):
                if not bits(outs[0][k].float(), outs[j][k].float()):
                    bad.append(f"draws={draws}: {k} differs between chunk {chunks[0]} and chunk {chunks[j]}")
        if not draws and static is None:
            # (the oracle on the rays this device computed: torch's own get_rays expression differs by an ulp between host
            # and device -- the 3-term sum is associated differently by the device's reduction kernel -- and the encoding's top frequency turns an ulp of a
            # direction into 1e-4 of a map; the first version of this campaign measured exactly that)
            o, dd = (t.cpu() for t in P.get_rays(H, W_, K, c2w))
            ref = orc.render_rays(orc.pack_ray_batch(o, dd, 2.0, 6.0), sds[0], sds[1], Ns, mode, "midpoint", perturb=0.0, N_importance=Ni,
                                  white_bkgd=kw["white_bkgd"], lindisp=kw["lindisp"])
            e = max(float(((outs[0][k].reshape(n_pix, -1).cpu().double() - ref[k].reshape(n_pix, -1).double()).abs()
                           / (1.0 + ref[k].reshape(n_pix, -1).double().abs())).max()) for k in ("rgb0", "acc0", "depth0"))
            stats["coarse_worst"] = max(stats["coarse_worst"], e)
            if e > 1e-5:
                bad.append(f"coarse maps vs the oracle {e:.2e}")
    stats["cases"] += 1
    stats["pixels"] += n_pix
    if bad:
        violations.append({"case": case, "H": H, "W": W_, "mode": mode, "precision": prec, "chunks": chunks, "what": bad[:6]})
print(json.dumps({"what": "render() chunk invariance, bit for bit, with and without counter-based draws", "seed": a.seed, "stats": stats,
                  "violations": violations}))
sys.exit(1 if violations else 0)
