"""MLP campaign: run_network (run_plnerf.py:78-92) = Embedder (run_nerf_helpers.py:24-54) + NeRF.forward (:105-128) and its
backward on random SUPPORTED architectures (netdepth 1-8, netwidth 8-256, skip position, with / without view directions,
multires 0-10, multires_views 0-4), ragged row counts and samples per ray, the fused entry (pts / viewdirs, in-kernel encoding)
and the embedded entry, in exact fp32 and f16x3 -- against the same network written out in fp64 torch.  Shapes the compiled
trunk cannot express (`is_supported()` false: a skip with too many layers before or behind it, ...) run on the layer-by-layer
route (generic.py: exact fp32 whatever the mode; until round 5 they were refused) through run_network, fp32 bounds.

Bounds: forward 1e-5 (fp32) / 1e-5 (f16x3: the contract) abs + rel on every row; every real parameter's gradient within
2e-4 of that tensor's max |g| (fp32) / 6e-3 of the network's largest gradient entry (f16x3: half planes) -- except on cases whose fp64 reference holds a ReLU unit within the mode's
forward error of zero (counted separately: such a unit takes the other side).
Test infrastructure.  python tools/fuzz_mlp.py --cases 200 --seed 21 > out.json"""
import argparse, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import plnerf_amd as P

ap = argparse.ArgumentParser()
ap.add_argument("--cases", type=int, default=200)
ap.add_argument("--seed", type=int, default=21)
a = ap.parse_args()
dev = torch.device("cuda:0")
rng = np.random.default_rng(a.seed)
F = torch.nn.functional
g = lambda x: x.to(dev)
FWD_TOL = {"fp32": 1e-5, "f16x3": 1e-5}
GRAD_TOL = {"fp32": 2e-4, "f16x3": 6e-3}
FLIP_BELOW = {"fp32": 2e-6, "f16x3": 1e-5}
stats = {"run": 0, "refused": 0, "generic": 0, "rows": 0, "forward_worst": {"fp32": 0.0, "f16x3": 0.0}, "grad_worst_no_near_zero_unit": {"fp32": 0.0, "f16x3": 0.0},
         "grad_worst_with_near_zero_unit": {"fp32": 0.0, "f16x3": 0.0}, "cases_with_near_zero_unit": {"fp32": 0, "f16x3": 0}}
violations, refused_shapes = [], []
for case in range(a.cases):
    D = int(rng.integers(1, 9))
    Wd = int(rng.choice([8, 32, 64, 96, 128, 200, 256]))
    skip = int(rng.choice([4, 4, 4, 0, 1, 2, 3]))
    use_vd = bool(rng.integers(4) != 0)
    L, M = int(rng.integers(0, 11)), int(rng.integers(0, 5))
    R, S = int(rng.choice([1, 3, 17, 64])), int(rng.choice([1, 5, 37, 64, 192]))
    torch.manual_seed(100 + case)
    emb_fn, in_ch = P.get_embedder(L, 0)
    embd_fn, in_v = P.get_embedder(M, 0)
    shape = {"D": D, "W": Wd, "skip": skip, "use_viewdirs": use_vd, "multires": L, "multires_views": M, "R": R, "S": S}
    nets = {}
    try:
        for prec in FWD_TOL:
            torch.manual_seed(100 + case)
            nets[prec] = P.NeRF(D=D, W=Wd, input_ch=in_ch, input_ch_views=in_v if use_vd else 0, output_ch=5, skips=[skip],
                                use_viewdirs=use_vd, precision=prec).to(dev)
    except NotImplementedError:
        stats["refused"] += 1; refused_shapes.append(shape); continue
    generic = not nets["fp32"].is_supported()
    if generic:
        if skip == D - 1:      # (a skip after the last trunk layer: the reference's own head cannot consume it)
            stats["refused"] += 1; refused_shapes.append(shape); continue
        stats["generic"] += 1; refused_shapes.append(shape)
        nets = {"fp32": nets["fp32"]}
    gen = torch.Generator().manual_seed(200 + case)
    pts = (torch.rand(R, S, 3, generator=gen) * 2 - 1) * 1.5
    vd = F.normalize(torch.randn(R, 3, generator=gen), dim=-1)
    cot = torch.randn(R, S, 4, generator=gen)
    sd = {k: v.detach().cpu().double().requires_grad_(True) for k, v in nets["fp32"].state_dict().items()}
    x = emb_fn(pts.reshape(-1, 3)).double()
    v = embd_fn(vd[:, None].expand(R, S, 3).reshape(-1, 3)).double()
    h, small = x, float("inf")
    for i in range(D):
        z = F.linear(h, sd[f"pts_linears.{i}.weight"], sd[f"pts_linears.{i}.bias"])
        small = min(small, float(z.detach().abs().min()))
        h = F.relu(z)
        if i == skip:
            h = torch.cat([x, h], -1)
    if use_vd:
        sigma = F.linear(h, sd["alpha_linear.weight"], sd["alpha_linear.bias"])
        feat = F.linear(h, sd["feature_linear.weight"], sd["feature_linear.bias"])
        zv = F.linear(torch.cat([feat, v], -1), sd["views_linears.0.weight"], sd["views_linears.0.bias"])
        small = min(small, float(zv.detach().abs().min()))
        ref = torch.cat([F.linear(F.relu(zv), sd["rgb_linear.weight"], sd["rgb_linear.bias"]), sigma], -1)
    else:
        ref = F.linear(h, sd["output_linear.weight"], sd["output_linear.bias"])[:, :4]
    (ref * cot.reshape(-1, 4).double()).sum().backward()
    stats["run"] += 1
    stats["rows"] += R * S
    for prec, net in nets.items():
        bad = []
        emb_in = (torch.cat([x, v], -1) if use_vd else x).float()
        if generic:      # (no fused entry: the reference's call, which embeds with plnerf_embed_rows and goes layer by layer)
            out = P.run_network(g(pts), g(vd) if use_vd else None, net, emb_fn, embd_fn)[..., :4]
        else:
            out = net.query(g(pts), g(vd) if use_vd else None)[..., :4]
        out_e = net(g(emb_in))[..., :4]
        r32 = ref.detach()
        e = max(float(((o.detach().cpu().double().reshape(-1, 4) - r32).abs() / (1.0 + r32.abs())).max()) for o in (out, out_e))
        stats["forward_worst"][prec] = max(stats["forward_worst"][prec], e)
        if e > FWD_TOL[prec]:
            bad.append(f"forward {e:.2e}")
        net.zero_grad()
        (out * g(cot)).sum().backward()
        # fp32: every tensor against ITS OWN largest entry.  f16x3: against the network's largest gradient entry -- the backward
        # runs on half planes, a bias gradient is a row sum of dz entries that may cancel to 1e-3 of their size, and 2^-11 of
        # the ENTRIES is then more than the sum (seed 22, case 50: a 15-row batch on an 8-wide network, 2.2e-4 absolute
        # = 1.06 of that bias's own 2.1e-4 = 3e-3 of the network's largest entry)
        pairs = []
        for name, prm in net.named_parameters():
            r = sd[name].grad
            if r is None or prm.grad is None:
                continue
            got = prm.grad.cpu().double()
            if name.startswith("output_linear"):
                r, got = r[:4], got[:4]
            pairs.append((name, got, r))
        g_max = max(float(r.abs().max()) for _, _, r in pairs)
        worst, which = 0.0, None
        for name, got, r in pairs:
            scale = max(float(r.abs().max()), 1e-9) if prec == "fp32" else max(g_max, 1e-9)
            ew = float((got - r).abs().max()) / scale
            if ew > worst:
                worst, which = ew, name
        near = small < FLIP_BELOW[prec]
        key = "grad_worst_with_near_zero_unit" if near else "grad_worst_no_near_zero_unit"
        stats[key][prec] = max(stats[key][prec], worst)
        stats["cases_with_near_zero_unit"][prec] += int(near)
        if worst > GRAD_TOL[prec] and not near:
            bad.append(f"gradient of {which}: {worst:.2e} of max |g| (smallest |pre-activation| {small:.1e})")
        if bad:
            violations.append({"case": case, "precision": prec, "shape": shape, "what": bad})
print(json.dumps({"what": "MLP forward / backward campaign over supported architectures vs fp64 torch", "seed": a.seed, "stats": stats,
                  "generic_or_refused_shapes_sample": refused_shapes[:8], "violations": violations}))
sys.exit(1 if violations else 0)
