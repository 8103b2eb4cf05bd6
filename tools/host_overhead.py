"""How far ahead of the GPU the host runs: wall time the host needs to ENQUEUE n steps of bench.py's step (no
synchronisation in between) against the time the GPU needs to execute them."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
sys.argv = [sys.argv[0]]
a = bench.parse()
ns, ni, _ = bench.WORKLOADS[a.workload]
a.n_samples, a.n_importance = ns, ni
import plnerf_amd as P
dev = torch.device("cuda", 0)
scene = bench.Scene(P, a.workload, a.views, dev)
step, nets = bench.build_step(P, a, a.precision, scene, dev, 0, 1, False)
for i in range(10):
    step(i)
torch.cuda.synchronize()
out = []
for n in (5, 20, 60):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n):
        step(100 + i)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    out.append({"steps": n, "host_enqueue_ms_per_step": 1e3 * (t1 - t0) / n, "total_ms_per_step": 1e3 * (t2 - t0) / n})
print(json.dumps(out))
