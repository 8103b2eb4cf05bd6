"""Per-tensor gradient error of a precision mode against the fp32 oracle's autograd, at a chosen batch size
(samples with a ReLU pre-activation within `amb` of zero are masked out of the cotangent, as in the tests).
    python tools/debug_grads.py f16x3 96 192"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
import plnerf_amd as P
from oracle import plnerf_oracle as orc
from test_gpu_parity import ambiguous_rows
dev = torch.device("cuda:0")
prec = sys.argv[1] if len(sys.argv) > 1 else "bf16x3"
R, S = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (5, 64)
gen = torch.Generator().manual_seed(R * 100 + S)
pts = (torch.rand(R, S, 3, generator=gen) * 2 - 1) * 2.5
vd = torch.nn.functional.normalize(torch.randn(R, 3, generator=gen), dim=-1)
cot = torch.randn(R, S, 4, generator=gen)
sd = orc.closed_form_state_dict(3, False)
keep = ~ambiguous_rows(sd, pts, vd, 5e-5)
cot = cot * keep.reshape(R, S, 1)
sd_o = {k: v.clone().double().requires_grad_(True) for k, v in sd.items()}
raw_o = orc.query_network(sd_o, pts.double(), vd.double())
(raw_o * cot.double()).sum().backward()
for pr in ("fp32", prec):
    net = P.NeRF(D=8, W=256, input_ch=63, input_ch_views=27, output_ch=5, skips=[4], use_viewdirs=True, precision=pr)
    net.load_state_dict(sd); net = net.to(dev)
    raw_h = net.query(pts.to(dev), vd.to(dev))
    (raw_h * cot.to(dev)).sum().backward()
    print("==", pr, f"({int(keep.sum())} of {keep.numel()} samples)")
    for name, prm in net.named_parameters():
        ref = sd_o[name].grad
        d = (prm.grad.cpu().double() - ref).abs()
        err = float(d.max())
        print(f"{name:28s} max|g| {float(ref.abs().max()):.3e} err {err:.3e} rel {err/max(float(ref.abs().max()),1e-12):.3e} "
              f"rms rel {float(d.pow(2).mean().sqrt())/max(float(ref.abs().max()),1e-12):.3e}")
