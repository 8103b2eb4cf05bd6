"""Which aten ops of bench.py's step launch small copies / fills (the __amd_rocclr_copyBuffer / fillBuffer rows of the kernel
stats), with the Python line that asked for them:   python tools/find_copies_in_step.py [--workload ...]"""
import os, sys, collections
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
for i in range(10):
    step(i)
torch.cuda.synchronize()
N = 4
with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CPU, torch.profiler.ProfilerActivity.CUDA],
                            with_stack=True) as prof:
    for i in range(N):
        step(100 + i)
    torch.cuda.synchronize()
count = collections.Counter()
for ev in prof.events():
    if ev.name.startswith("aten::") and ev.name.split("::")[1] in ("copy_", "fill_", "zero_", "zeros", "ones", "full", "clone",
                                                                 "contiguous", "to", "_to_copy", "cat", "stack", "mul", "add",
                                                                 "div", "sub", "neg", "sum", "mean", "log", "pow", "select",
                                                                 "index", "empty_strided") and ev.cpu_parent is None:
        where = next((s for s in (ev.stack or []) if "/repo/" in s and "tools/" not in s), "?")
        count[(ev.name, where[-90:])] += 1
for (name, where), c in sorted(count.items(), key=lambda kv: -kv[1]):
    print(f"{c / N:5.1f} per step  {name:22s} {where}")
print()
k = collections.Counter()
for ev in prof.events():
    if ev.device_type == torch.autograd.DeviceType.CUDA:
        k[ev.name[:60]] += 1
for name, c in sorted(k.items(), key=lambda kv: -kv[1])[:45]:
    print(f"{c / N:5.1f} per step  {name}")
