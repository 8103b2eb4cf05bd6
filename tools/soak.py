"""Soak run of the training step: N steps of bench.py's own step (device-side pixel choice .. Adam), reporting loss
finiteness, step-time drift, the range status word, and whether device memory grows (allocator high-water marks at
10 % and 100 % of the run)."""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=3000)
ap.add_argument("--precision", default="f16x3")
ap.add_argument("--workload", default="blender_64_128")
sa = ap.parse_args()
sys.argv = [sys.argv[0], "--precision", sa.precision, "--workload", sa.workload]
a = bench.parse()
ns, ni, _ = bench.WORKLOADS[a.workload]
a.n_samples, a.n_importance = ns, ni
import plnerf_amd as P
dev = torch.device("cuda", 0)
scene = bench.Scene(P, a.workload, a.views, dev)
step, nets = bench.build_step(P, a, a.precision, scene, dev, 0, 1, False)
losses, marks, times = [], {}, []
block = max(sa.steps // 10, 1)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(sa.steps):
    loss = step(i)
    if i % block == block - 1:
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        times.append(1e3 * (t1 - t0) / block)
        losses.append(float(loss))
        marks[i + 1] = torch.cuda.max_memory_allocated() / 2**30
        t0 = time.perf_counter()
status = [int(n.range_status()) if hasattr(n, "range_status") else 0 for n in nets]
print(json.dumps({"what": f"soak, {sa.workload}, {sa.precision}", "steps": sa.steps,
                  "ms_per_step_by_tenth": [round(t, 3) for t in times],
                  "loss_by_tenth": [round(l, 6) for l in losses], "all_finite": all(l == l and abs(l) < 1e30 for l in losses),
                  "max_memory_allocated_GiB_by_tenth": {k: round(v, 3) for k, v in marks.items()},
                  "range_status_words": status}))
