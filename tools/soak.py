"""Soak run of the training step: N steps of bench.py's configuration, reporting loss finiteness, step-time drift and
whether device memory grows (allocator high-water marks at 10 % and 100 % of the run)."""
import argparse, json, os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import plnerf_amd as P
import bench

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=2000)
ap.add_argument("--precision", default="f16x3")
a = ap.parse_args()
dev = torch.device("cuda:0")
ck = tempfile.mkdtemp(); os.makedirs(os.path.join(ck, "exp"))
args = bench.make_args(argparse.Namespace(precision=a.precision, n_samples=64, n_importance=128, rays=4096), ck)
torch.manual_seed(0)
so = sys.stdout; sys.stdout = open(os.devnull, "w")
kw, _, _, _, opt, opt_c = P.create_nerf(args, device=dev)
sys.stdout = so
batch, target, K = P.rays.synthetic_blender_rays(4096, seed=0, device="cpu")
rays = (batch[0].to(dev), batch[1].to(dev)); target = target.to(dev)
losses, marks, times = [], {}, []
t0 = time.perf_counter()
for i in range(a.steps):
    rgb, disp, acc, extras = P.render(800, 800, K, chunk=32768, rays=rays, near=2.0, far=6.0, retraw=True, **kw)
    opt.zero_grad(); opt_c.zero_grad()
    loss = P.img2mse(rgb, target) + P.img2mse(extras["rgb0"], target)
    loss.backward(); opt.step(); opt_c.step()
    if i % 100 == 99 or i == 0:
        torch.cuda.synchronize()
        losses.append(float(loss.detach())); times.append(time.perf_counter() - t0)
    if i in (a.steps // 10, a.steps - 1):
        marks[i] = torch.cuda.max_memory_allocated() / 1e9
torch.cuda.synchronize()
per = [(times[k + 1] - times[k]) / 100 * 1e3 for k in range(1, len(times) - 1)]
print(json.dumps({"precision": a.precision, "steps": a.steps, "loss_first": losses[0], "loss_last": losses[-1],
                  "all_finite": all(l == l and abs(l) < 1e9 for l in losses),
                  "ms_per_step_first_100s": round(per[0], 3), "ms_per_step_last_100s": round(per[-1], 3),
                  "max_mem_GB": {str(k): round(v, 3) for k, v in marks.items()}}))
