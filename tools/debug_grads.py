import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import plnerf_amd as P
from oracle import plnerf_oracle as orc
dev = torch.device("cuda:0")
prec = sys.argv[1] if len(sys.argv) > 1 else "bf16x3"
R, S = 5, 64
gen = torch.Generator().manual_seed(R * 100 + S)
pts = (torch.rand(R, S, 3, generator=gen) * 2 - 1) * 2.5
vd = torch.nn.functional.normalize(torch.randn(R, 3, generator=gen), dim=-1)
cot = torch.randn(R, S, 4, generator=gen)
sd = orc.closed_form_state_dict(3, False)
sd_o = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
raw_o = orc.query_network(sd_o, pts, vd)
(raw_o * cot).sum().backward()
for pr in ("fp32", prec):
    net = P.NeRF(D=8, W=256, input_ch=63, input_ch_views=27, output_ch=5, skips=[4], use_viewdirs=True, precision=pr)
    net.load_state_dict(sd); net = net.to(dev)
    raw_h = net.query(pts.to(dev), vd.to(dev))
    (raw_h * cot.to(dev)).sum().backward()
    print("==", pr)
    for name, prm in net.named_parameters():
        ref = sd_o[name].grad
        err = float((prm.grad.cpu() - ref).abs().max())
        print(f"{name:28s} max|g| {float(ref.abs().max()):.3e} err {err:.3e} rel {err/max(float(ref.abs().max()),1e-12):.3e}")
