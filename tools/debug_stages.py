"""Stage-by-stage comparison of render_rays (HIP) against the oracle on identical draws."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import plnerf_amd as P
from plnerf_amd import functional as Fn
from oracle import plnerf_oracle as orc

dev = torch.device("cuda:0")
g = lambda x: x.to(dev)
md = lambda a, b: float((a.detach().cpu().double() - b.detach().cpu().double()).abs().max())

R, Ns, Ni = 12, 64, 128
batch, _ = orc.synthetic_blender_rays(R, seed=3)
sd_c, sd_f = orc.closed_form_state_dict(0, True), orc.closed_form_state_dict(1, True)
ref, it = orc.render_rays(batch, sd_c, sd_f, Ns, "linear", "midpoint", retraw=True, perturb=1.0, N_importance=Ni,
                          white_bkgd=True, pytest=True, return_internals=True)

def net(sd):
    n = P.NeRF(D=8, W=256, input_ch=63, input_ch_views=27, output_ch=5, skips=[4], use_viewdirs=True)
    n.load_state_dict(sd)
    return n.to(dev)
nc, nf = net(sd_c), net(sd_f)
rb = g(batch)
rays_o, rays_d, vd = rb[:, 0:3], rb[:, 3:6], rb[:, -3:]
near, far = rb[:, 6:7], rb[:, 7:8]
with torch.no_grad():
    # feed the ORACLE's intermediate inputs to each HIP stage
    z0 = g(it["z_coarse"])
    pts = rays_o[..., None, :] + rays_d[..., None, :] * z0[..., :, None]
    raw0 = nc.query(pts, vd)
    print("raw coarse", md(raw0, it["raw_coarse"]))
    out0 = P.raw2outputs(g(it["raw_coarse"]), z0, near, far, rays_d, "linear", "midpoint", white_bkgd=True)
    print("weights coarse", md(out0[3], it["weights_coarse"]))
    r_o = orc.raw2outputs(it["raw_coarse"], it["z_coarse"], batch[:, 6:7], batch[:, 7:8], batch[:, 3:6], "linear", "midpoint", white_bkgd=True)
    zs = P.sample_pdf_reformulation(z0, g(r_o[3]), g(r_o[5]), g(r_o[6]), near, far, Ni, det=False, pytest=True)[0]
    zs_o = torch.clamp(orc.sample_pdf_reformulation(it["z_coarse"], r_o[3], r_o[5], r_o[6], batch[:, 6:7], batch[:, 7:8], Ni, pytest=True)[0], batch[:, 6:7], batch[:, 7:8])
    print("z_samples (unclamped hip vs clamped oracle)", md(torch.clamp(zs, near, far), zs_o), " oracle internals:", md(zs_o, it["z_samples"]))
    zf = Fn.merge_sort(z0, g(it["z_samples"]), near, far)
    print("z fine", md(zf, it["z_fine"]))
    ptsf = rays_o[..., None, :] + rays_d[..., None, :] * g(it["z_fine"])[..., :, None]
    rawf = nf.query(ptsf, vd)
    rawf_o = orc.query_network(sd_f, (batch[:, 0:3][..., None, :] + batch[:, 3:6][..., None, :] * it["z_fine"][..., :, None]), batch[:, -3:])
    print("raw fine", md(rawf, rawf_o), "max|raw|", float(rawf_o.abs().max()))
    outf = P.raw2outputs(g(rawf_o), g(it["z_fine"]), near, far, rays_d, "linear", "midpoint", white_bkgd=True)
    print("weights fine", md(outf[3], it["weights_fine"]), "rgb", md(outf[0], ref["rgb_map"]))
    # now the composed call
    emb, _ = P.get_embedder(10, 0); embd, _ = P.get_embedder(4, 0)
    qfn = lambda inputs, viewdirs, fn: P.run_network(inputs, viewdirs, fn, emb, embd)
    got = P.render_rays(rb, nc, qfn, Ns, "linear", "midpoint", retraw=True, perturb=1.0, N_importance=Ni,
                        network_fine=nf, white_bkgd=True, pytest=True)
    for k in ref:
        print("composed", k, md(got[k], ref[k]))
