"""Upper bound of what ONE grid for both networks' backward could buy (VERDICT r04 #3a), without building it: the
backward of a single network over 786,432 + 262,144 = 1,048,576 rows IS the merged grid's best case (same kernels, same
tiles, one launch of each instead of two) -- compare it with the sum of the two separate backwards, interleaved on one
box.  Also the forward, for the same question.

    gpurun --timeout 600 -- 'python tools/merged_bwd_bound.py > gpurun_out/merged_bwd_bound.txt'
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import plnerf_amd as P
from plnerf_amd import functional as Fn

dev = torch.device("cuda:0")
torch.manual_seed(0)
net = P.NeRF(D=8, W=256, input_ch=63, input_ch_views=27, output_ch=5, skips=[4], use_viewdirs=True,
             precision=sys.argv[1] if len(sys.argv) > 1 else "f16x3").to(dev)
R = 4096
vd = torch.nn.functional.normalize(torch.randn(R, 3, device=dev), dim=-1)
cases = {}
for S in (192, 64, 256):
    pts = (torch.rand(R, S, 3, device=dev) * 2 - 1) * 3
    g = torch.randn(R, S, 4, device=dev) * 1e-3
    cases[S] = (pts, g)


def run(S, reps):
    pts, g = cases[S]
    timer = Fn.KernelTimer()
    Fn.KERNEL_TIMER = timer
    try:
        for _ in range(reps):
            net.zero_grad()
            raw = net.query(pts, vd)
            raw.backward(g)
        torch.cuda.synchronize()
    finally:
        Fn.KERNEL_TIMER = None
    rows = R * S
    return timer.mean_ms(f"mlp_fwd[{rows}]"), timer.mean_ms(f"mlp_bwd[{rows}]")


for S in cases:
    run(S, 3)      # warm-up
print(f"# {net.precision}: backward (absmax + dgrad + wgrad main / thin / head + reduce) and training forward of ONE network, "
      f"HIP events around plnerf_mlp_bwd / plnerf_mlp_fwd, mean of 10 launches per leg, legs interleaved")
for rnd in range(4):
    f192, b192 = run(192, 10)
    f64, b64 = run(64, 10)
    f256, b256 = run(256, 10)
    print(f"round {rnd}: bwd 786432 rows {b192:.4f} ms + 262144 rows {b64:.4f} ms = {b192 + b64:.4f} ms | 1048576 rows in one "
          f"launch sequence {b256:.4f} ms | merged would save at most {b192 + b64 - b256:+.4f} ms ; "
          f"fwd {f192:.4f} + {f64:.4f} = {f192 + f64:.4f} | {f256:.4f} ({f192 + f64 - f256:+.4f})")
