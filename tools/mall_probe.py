"""Does data a kernel has just written come back from the 256 MB Infinity Cache (MALL) rather than HBM?
For each buffer size: (write -> read) pairs, timing only the reads (sum) that directly follow a write
(mul_) of the same buffer, against reads of a buffer that was evicted by streaming 8 GiB in between."""
import json, torch
dev = "cuda"
big = torch.empty(2 << 30, device=dev, dtype=torch.float32)   # 8 GiB evictor
def ev(): return torch.cuda.Event(enable_timing=True)
for mb in (32, 64, 96, 128, 192, 256, 384, 512, 1024, 4096):
    n = mb * (1 << 20) // 4
    x = torch.randn(n, device=dev)
    warm, cold, wr = [], [], []
    for rep in range(6):
        a, b, c = ev(), ev(), ev()
        a.record(); x.mul_(1.0001); b.record(); s = x.sum(); c.record(); torch.cuda.synchronize()
        wr.append(a.elapsed_time(b)); warm.append(b.elapsed_time(c))
        big.fill_(0.0)
        a, b = ev(), ev()
        a.record(); s = x.sum(); b.record(); torch.cuda.synchronize()
        cold.append(a.elapsed_time(b))
    f = lambda t: mb * (1 << 20) / (sorted(t)[len(t) // 2] * 1e-3) / 1e9
    print(json.dumps({"MiB": mb, "read_after_write_GBps": round(f(warm)), "read_after_evict_GBps": round(f(cold)),
                      "rmw_GBps(2x bytes)": round(2 * f(wr))}))
