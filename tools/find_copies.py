"""Which host-side operations of one training step end in a runtime copy / fill launch (__amd_rocclr_copyBuffer,
fillBuffer)?  Runs bench.py's step under torch.profiler and prints, for every Memcpy / Memset / non-library kernel of
ONE step, the enclosing CPU operator chain."""
import os, sys, json, argparse
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("PLNERF_ALLOW_TOOLS_BUILD", "1")
import torch
from torch.profiler import profile, ProfilerActivity
import bench

sys.argv = [sys.argv[0]] + sys.argv[1:]
a = bench.parse()
ns, ni, _ = bench.WORKLOADS[a.workload]
a.n_samples = a.n_samples if a.n_samples is not None else ns
a.n_importance = a.n_importance if a.n_importance is not None else ni
import plnerf_amd as P
dev = torch.device("cuda", 0)
scene = bench.Scene(P, a.workload, a.views, dev)
step, nets = bench.build_step(P, a, a.precision, scene, dev, 0, 1, False)
for i in range(3):
    step(i)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step(3)
    torch.cuda.synchronize()
out = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out", "find_copies_trace.json")
os.makedirs(os.path.dirname(out), exist_ok=True)
prof.export_chrome_trace(out)
ev = json.load(open(out))["traceEvents"]
cpu = [e for e in ev if e.get("ph") == "X" and e.get("cat") in ("cpu_op", "user_annotation", "python_function")]
rt = {e["args"].get("correlation"): e for e in ev if e.get("ph") == "X" and e.get("cat") in ("cuda_runtime", "cuda_driver")
      and "args" in e}
gpu = [e for e in ev if e.get("ph") == "X" and e.get("cat") in ("kernel", "gpu_memcpy", "gpu_memset")]
print(f"{len(gpu)} device activities in one step")
for g in sorted(gpu, key=lambda e: e["ts"]):
    name = g["name"]
    lib = any(k in name for k in ("mlp_", "wgrad", "quad_", "coarse_", "select_rays", "image_loss", "adam_kernel", "absmax",
                                  "pack_", "rr_pack", "sample_", "merge_sort"))
    r = rt.get(g.get("args", {}).get("correlation"))
    chain = []
    if r is not None:
        t = r["ts"]
        enc = [c for c in cpu if c["ts"] <= t <= c["ts"] + c["dur"] and c.get("tid") == r.get("tid")]
        enc.sort(key=lambda c: c["dur"])
        chain = [c["name"] for c in enc[:6]]
    tag = "   " if lib else ">>>"
    print(f"{tag} {g['cat']:11s} {g['dur']:8.1f} us  {name[:70]:70s} <- {(r or {}).get('name', '?')} | {' < '.join(chain)[:400]}")
