"""MLP-only benchmark: the fused PE+MLP forward at the north-star batch (65536 x 192 samples by
default), per precision mode, inference (no saved activations) and training forward, plus the
dgrad + wgrad backward.  Prints one JSON line per configuration."""
import argparse, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import plnerf_amd as P

FWD, TRAIN = 1186816, 3489024
PEAK = {"fp32": 157.3, "bf16x3": 2500.0, "bf16": 2500.0, "f16x3": 2500.0, "f16": 2500.0}
ap = argparse.ArgumentParser()
ap.add_argument("--rays", type=int, default=65536)
ap.add_argument("--samples", type=int, default=192)
ap.add_argument("--precisions", default="fp32,f16x3,bf16x3,f16,bf16")
ap.add_argument("--iters", type=int, default=5)
ap.add_argument("--train-rays", type=int, default=8192)
a = ap.parse_args()
dev = torch.device("cuda:0")
torch.manual_seed(0)


def timeit(fn, iters):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


for prec in a.precisions.split(","):
    net = P.NeRF(D=8, W=256, input_ch=63, input_ch_views=27, output_ch=5, skips=[4], use_viewdirs=True,
                 precision=prec).to(dev)
    R, S = a.rays, a.samples
    pts = (torch.rand(R, S, 3, device=dev) * 2 - 1) * 3
    vd = torch.nn.functional.normalize(torch.randn(R, 3, device=dev), dim=-1)
    with torch.no_grad():
        ms = timeit(lambda: net.query(pts, vd), a.iters)
    rows = R * S
    tf = rows * FWD / (ms * 1e-3) / 1e12
    print(json.dumps({"what": "fused MLP forward (inference)", "precision": prec, "rows": rows, "ms": ms,
                      "tflops": tf, "peak": PEAK[prec], "frac": tf / PEAK[prec]}), flush=True)
    Rt = a.train_rays
    if Rt <= 0:
        del net, pts, vd
        torch.cuda.empty_cache()
        continue
    ptst, vdt = pts[:Rt].contiguous(), vd[:Rt].contiguous()
    cot = torch.randn(Rt, S, 4, device=dev)

    def train():
        for p in net.parameters():
            p.grad = None
        (net.query(ptst, vdt) * cot).sum().backward()
    ms = timeit(train, max(2, a.iters // 2))
    rows = Rt * S
    tf = rows * TRAIN / (ms * 1e-3) / 1e12
    print(json.dumps({"what": "MLP forward+backward (training)", "precision": prec, "rows": rows, "ms": ms,
                      "tflops": tf}), flush=True)
    del net, pts, vd
    torch.cuda.empty_cache()
