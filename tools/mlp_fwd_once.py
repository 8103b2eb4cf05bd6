"""Run the fused MLP forward (inference) a few times for one precision -- a small target for
rocprofv3 --pmc passes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import plnerf_amd as P
prec = sys.argv[1] if len(sys.argv) > 1 else "bf16"
R = int(sys.argv[2]) if len(sys.argv) > 2 else 16384
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 3
dev = torch.device("cuda:0")
torch.manual_seed(0)
net = P.NeRF(D=8, W=256, input_ch=63, input_ch_views=27, output_ch=5, skips=[4], use_viewdirs=True, precision=prec).to(dev)
pts = (torch.rand(R, 192, 3, device=dev) * 2 - 1) * 3
vd = torch.nn.functional.normalize(torch.randn(R, 3, device=dev), dim=-1)
with torch.no_grad():
    for _ in range(iters):
        net.query(pts, vd)
torch.cuda.synchronize()
