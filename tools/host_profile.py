"""Where the host's time per step goes (cProfile over an un-synchronised loop of bench.py's step at a batch small enough
to be host-bound):   python tools/host_profile.py [--rays 512]"""
import cProfile, io, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
a = bench.parse()
ns, ni, _ = bench.WORKLOADS[a.workload]
a.n_samples, a.n_importance = ns, ni
import plnerf_amd as P
dev = torch.device("cuda", 0)
scene = bench.Scene(P, a.workload, a.views, dev)
step, nets = bench.build_step(P, a, a.precision, scene, dev, 0, 1, False)
for i in range(20):
    step(i)
torch.cuda.synchronize()
n = 200
t0 = time.perf_counter()
for i in range(n):
    step(100 + i)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"rays {a.rays}: host enqueue {1e3 * (t1 - t0) / n:.3f} ms/step, total {1e3 * (t2 - t0) / n:.3f} ms/step")
pr = cProfile.Profile()
pr.enable()
for i in range(n):
    step(1000 + i)
pr.disable()
torch.cuda.synchronize()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(45)
print("\n".join(l[:150] for l in s.getvalue().splitlines()))
