"""Bare latency of the step's collective on this node: torch.distributed.all_reduce (RCCL, SUM, in place) of one network's
flat gradient buffer -- 595,844 + 4 floats = 2.38 MB -- and of both networks' at once, 100 times each, HIP events on the
launch stream.  One rank per GPU from torchrun's environment; rank 0 prints one JSON line.  (tools/scale_run.py)"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist
from plnerf_amd import dp

rank, world, local = dp.init_from_env(force=True)
dev = torch.device("cuda", local)
out = {"world": world, "backend": dist.get_backend()}
for name, n in (("one_network_2.38MB", 595848), ("both_networks_4.77MB", 2 * 595848)):
    x = torch.ones(n, device=dev)
    for _ in range(10):
        dist.all_reduce(x)
    torch.cuda.synchronize()
    dist.barrier(device_ids=[local])
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(101)]
    evs[0].record()
    for i in range(100):
        dist.all_reduce(x)
        evs[i + 1].record()
    torch.cuda.synchronize()
    us = sorted(1e3 * evs[i].elapsed_time(evs[i + 1]) for i in range(100))
    t = torch.tensor([us[50]], device=dev, dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    out[name] = {"floats": n, "us_median_max_over_ranks": float(t.item()), "us_min_rank0": us[0], "us_p90_rank0": us[90]}
if rank == 0:
    print(json.dumps(out), flush=True)
dist.barrier(device_ids=[local])
dist.destroy_process_group()
