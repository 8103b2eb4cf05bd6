"""HBM-bound kernels of the path at a size that fills the chip (R rays): achieved GB/s on the ALGORITHMIC bytes of
SURVEY section 8d (quadrature 32 S + 64 B/ray forward, 52 S backward; PL sampler 4 (4 S + 7 + N) + 4 N; constant
sampler 4 (2 B + N) + 12 N; merge sort 8 (S + N); sampling prologue 8 S and 16 S), against the 6.26 TB/s a pure reader
sustains on this part (profiles/r01_hbm_read_probe.txt) and the 8 TB/s datasheet figure."""
import argparse, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import plnerf_amd as P
from plnerf_amd import functional as Fn

ap = argparse.ArgumentParser()
ap.add_argument("--rays", type=int, default=262144)
ap.add_argument("--reps", type=int, default=10)
a = ap.parse_args()
dev = torch.device("cuda:0")
R, S, N = a.rays, 192, 128
g = torch.Generator(device=dev).manual_seed(0)
rnd = lambda *s: torch.rand(*s, device=dev, generator=g)
near, far = torch.full((R, 1), 2.0, device=dev), torch.full((R, 1), 6.0, device=dev)
rays_o, rays_d = rnd(R, 3), torch.nn.functional.normalize(rnd(R, 3) - 0.5, dim=-1)

def timeit(f, reps=None):
    reps = reps or a.reps
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps

out = []
def report(name, ms, bytes_per_ray):
    gbs = R * bytes_per_ray / (ms * 1e-3) / 1e9
    out.append({"kernel": name, "rays": R, "ms": round(ms, 4), "bytes_per_ray": bytes_per_ray, "GBps": round(gbs),
                "frac_of_8TBps": round(gbs / 8000, 3), "frac_of_read_ceiling": round(gbs / 6260, 3)})

with torch.no_grad():
    t_vals = Fn.cpu_linspace(64, dev); t_rand = rnd(R, 64)
    report("stratified_z (S=64)", timeit(lambda: Fn.stratified_z(near, far, t_vals, t_rand)), 8 * 64 + 8)
    z64 = Fn.stratified_z(near, far, t_vals, t_rand)
    z = torch.sort(2.0 + 4.0 * rnd(R, S), dim=-1).values
    report("ray_points (S=192)", timeit(lambda: Fn.ray_points(rays_o, rays_d, z)), 16 * S + 24)
    raw = torch.randn(R, S, 4, device=dev, generator=g)
    report("quad_fwd linear (S=192)", timeit(lambda: P.raw2outputs(raw, z, near, far, rays_d, "linear", "midpoint", white_bkgd=True)), 32 * S + 64)
    raw64 = torch.randn(R, 64, 4, device=dev, generator=g)
    rgb, disp, acc, w, depth, tau, T = P.raw2outputs(raw64, z64, near, far, rays_d, "linear", "midpoint", white_bkgd=True)
    report("sample_pl (S=64, N=128)", timeit(lambda: P.sample_pdf_reformulation(z64, w, tau, T, near, far, N, det=False)), 4 * (4 * 64 + 7 + N) + 4 * N)
    zs = P.sample_pdf_reformulation(z64, w, tau, T, near, far, N, det=False)[0]
    report("merge_sort (64 + 128)", timeit(lambda: Fn.merge_sort(z64, zs, near, far)), 8 * (64 + N))
    zmid = .5 * (z64[..., 1:] + z64[..., :-1])
    wc = rnd(R, 62)
    report("sample_const (B=63, N=128)", timeit(lambda: P.sample_pdf(zmid, wc, N, det=False)), 4 * (2 * 63 + N) + 12 * N)
    # round 2: the fused step-level kernels (csrc/epilogue.hip, csrc/step.hip)
    u = rnd(R, N)
    rays_o3 = rays_o
    fused = lambda: Fn.CoarseEpilogueFn.apply(raw64, z64, near, far, rays_o3, rays_d, None, u, N, "midpoint", True, False,
                                              1e-4, 1e-3, None)
    # in: raw 16 S, z 4 S, u 4 N, near/far/o/d 32; out: maps 28, z_std 4, z_fine 4 (S+N), pts 12 (S+N)
    report("coarse_epilogue fused (S=64, N=128): quad + sample_pl + clamp + sort + points + z_std",
           timeit(fused), 20 * 64 + 4 * N + 32 + 32 + 16 * (64 + N))
    src = Fn.DrawSource(seed=1)
    fused_rng = lambda: Fn.CoarseEpilogueFn.apply(raw64, z64, near, far, rays_o3, rays_d, None, None, N, "midpoint", True,
                                                  False, 1e-4, 1e-3, src)
    report("coarse_epilogue fused, draws in the kernel (S=64, N=128)", timeit(fused_rng), 20 * 64 + 32 + 32 + 16 * (64 + N))
    report("coarse_samples fused (S=64): stratified z + points, draws in the kernel",
           timeit(lambda: Fn.coarse_samples(rays_o3, rays_d, near, far, t_vals, None, False, True, src)), 32 + 16 * 64)
    sep = lambda: (P.raw2outputs(raw64, z64, near, far, rays_d, "linear", "midpoint", white_bkgd=True),
                   Fn.ray_points(rays_o3, rays_d, Fn.merge_sort(z64, P.sample_pdf_reformulation(z64, w, tau, T, near, far, N, det=False)[0], near, far)))
    report("the same as separate launches (quad_fwd, rand, sample_pl, merge_sort, ray_points)", timeit(sep),
           20 * 64 + 4 * N + 32 + 32 + 16 * (64 + N))
raw.requires_grad_(True)
res = P.raw2outputs(raw, z, near, far, rays_d, "linear", "midpoint", white_bkgd=True)
gr = torch.randn_like(res[0])
def bwd():
    raw.grad = None
    res[0].backward(gr, retain_graph=True)
report("quad_bwd linear (S=192)", timeit(bwd), 52 * S)
for o in out:
    print(json.dumps(o))
