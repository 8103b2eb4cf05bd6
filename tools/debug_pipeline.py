"""Which parameters differ between TrainStep pipelines, and after which step (debug aid for
tests/test_gpu_step.py::test_two_stream_step_equals_one_stream_step)."""
import os, sys, tempfile
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import plnerf_amd as P
from test_gpu_step import _nets
dev = torch.device("cuda:0")
H = W = 160
K = [[220.0, 0, W / 2], [0, 220.0, H / 2], [0, 0, 1]]
gen = torch.Generator().manual_seed(5)
image = torch.rand(H, W, 3, generator=gen).to(dev)
poses = [P.rays.pose_spherical(-180.0 + 72.0 * i, -30.0, 4.0)[:3, :4] for i in range(5)]


def run(pipeline, own_rays, nsteps, **over):
    args, kw, opt, opt_c = _nets(P, **over)
    ts = P.TrainStep(args, kw, opt, opt_c, distributed=False, seed=17, pipeline=pipeline)
    snaps = []
    for step in range(nsteps):
        if own_rays:
            cols, target, _ = P.select_view_rays(H, W, K, poses[step % 5], image, 2048, 2.0, 6.0, seed=17, step=step)
            loss, psnr = ts(H, W, K, cols, target, near=2.0, far=6.0)
        else:
            loss, psnr = ts.step_view(H, W, K, poses[step % 5], image, near=2.0, far=6.0, n_rand=2048)
        ts.drain(); torch.cuda.synchronize()
        snaps.append((float(loss), [p.detach().clone() for n in ts.nets for p in n.parameters()],
                      [p.grad.detach().clone() for n in ts.nets for p in n.parameters()]))
    return snaps, [n for net in ts.nets for n, _ in net.named_parameters()]


for own_rays, over in ((False, {}), (True, {}), (False, {"constant_init": 3})):
    base, names = run(0, own_rays, 4, **over)
    for pipeline in (1, 2):
        got, _ = run(pipeline, own_rays, 4, **over)
        for step, ((l0, p0, g0), (l1, p1, g1)) in enumerate(zip(base, got)):
            bad_p = [(i, names[i], float((a - b).abs().max())) for i, (a, b) in enumerate(zip(p0, p1)) if not torch.equal(a, b)]
            bad_g = [(i, names[i], float((a - b).abs().max())) for i, (a, b) in enumerate(zip(g0, g1)) if not torch.equal(a, b)]
            print(f"own_rays={own_rays} over={over} pipeline={pipeline} step={step}: loss {l0 == l1}  params differing {len(bad_p)} grads differing {len(bad_g)}",
                  bad_p[:3], bad_g[:3])
