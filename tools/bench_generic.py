import sys, time, torch
sys.path.insert(0, "/root/repo")
import plnerf_amd as P
dev = torch.device("cuda:0")
emb, ic = P.get_embedder(10, 0); embd, icv = P.get_embedder(4, 0)
R, S = 4096, 64
pts = (torch.rand(R, S, 3, device=dev) * 2 - 1) * 2
vd = torch.nn.functional.normalize(torch.randn(R, 3, device=dev), dim=-1)
for name, kw, prec in (("generic 8x512", dict(D=8, W=512), "fp32"), ("generic 10x256 (skips 4,7)", dict(D=10, W=256, skips=[4, 7]), "fp32"),
                       ("fused fp32 8x256", dict(D=8, W=256), "fp32"), ("fused f16x3 8x256", dict(D=8, W=256), "f16x3")):
    net = P.NeRF(input_ch=ic, input_ch_views=icv, output_ch=5, use_viewdirs=True, precision=prec, **kw).to(dev)
    for it in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        out = P.run_network(pts, vd, net, emb, embd)
        torch.cuda.synchronize(); t1 = time.perf_counter()
        out[..., :4].sum().backward()
        torch.cuda.synchronize(); t2 = time.perf_counter()
    macs = sum(p.numel() for n, p in net.named_parameters() if n.endswith("weight"))
    print(f"{name}: {R*S} rows, forward {1e3*(t1-t0):.2f} ms ({2*macs*R*S/(t1-t0)/1e12:.1f} TFLOP/s), backward {1e3*(t2-t1):.2f} ms, supported={net.is_supported()}")
