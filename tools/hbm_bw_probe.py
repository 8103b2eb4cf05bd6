"""HBM bandwidth probe with plain torch ops (fill = write-only, sum = read-only, copy = 1R+1W) on
a buffer far larger than L2 + MALL.  Prints GB/s per op; context for the plane-writing kernels."""
import torch, json
n = 2 << 30  # fp32 elements = 8 GiB
x = torch.empty(n, device="cuda", dtype=torch.float32)
y = torch.empty(n, device="cuda", dtype=torch.float32)
def timeit(f, reps=5):
    f(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps
out = {}
ms = timeit(lambda: x.fill_(1.0)); out["fill_write_only_GBps"] = 4 * n / ms / 1e6
ms = timeit(lambda: x.sum()); out["sum_read_only_GBps"] = 4 * n / ms / 1e6
ms = timeit(lambda: y.copy_(x)); out["copy_1r1w_GBps"] = 8 * n / ms / 1e6
print(json.dumps(out))
