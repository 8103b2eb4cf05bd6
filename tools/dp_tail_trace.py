"""What sits between the backward's last kernel and Adam under data parallelism: two ranks (gloo, both on
cuda:0 -- RCCL refuses two ranks on one device) run the real TrainStep; rank 0 records one step with torch.profiler and
prints its GPU kernels in launch order, then the launches between each network's wgrad_reduce_kernel (the backward's
last kernel) and its adam_kernel.  With the exchange carrying the range status as a tail element and the 1 / world factor
inside the step kernel, that list holds only gloo's own device <-> host copies (which RCCL does not have): no scaling
launch, no status collective.
    python tools/dp_tail_trace.py            (spawns the two ranks itself)"""
import os, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))

if "RANK" not in os.environ:
    port = 29950 + os.getpid() % 40
    procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:],
                              env=dict(os.environ, RANK=str(r), WORLD_SIZE="2", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1",
                                       MASTER_PORT=str(port))) for r in range(2)]
    sys.exit(max(p.wait() for p in procs))

import torch
import plnerf_amd as P
from plnerf_amd import dp
from oracle import plnerf_oracle as orc
from test_gpu_step import _args

rank, world, _ = dp.init_from_env(backend="gloo")
dev = torch.device("cuda:0")
torch.cuda.set_device(0)
d = tempfile.mkdtemp(); os.makedirs(os.path.join(d, "exp"))
args = _args(d, "f16x3")
kw, _, _, _, opt, opt_c = P.create_nerf(args, device=dev)
kw["network_fn"].load_state_dict(orc.closed_form_state_dict(0, False))
kw["network_fine"].load_state_dict(orc.closed_form_state_dict(1, False))
ts = P.TrainStep(args, kw, opt, opt_c, distributed=True, seed=3)
H = W = 128
K = [[180.0, 0, W / 2], [0, 180.0, H / 2], [0, 0, 1]]
c2w = P.rays.pose_spherical(20.0, -30.0, 4.0)[:3, :4]
image = torch.rand(H, W, 3, generator=torch.Generator().manual_seed(1)).to(dev)
for _ in range(3):
    ts.step_view(H, W, K, c2w, image, near=2.0, far=6.0, n_rand=2048)
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    ts.step_view(H, W, K, c2w, image, near=2.0, far=6.0, n_rand=2048)
    torch.cuda.synchronize()
if rank == 0:
    evs = [e for e in prof.events() if e.device_type == torch.autograd.DeviceType.CUDA]
    evs.sort(key=lambda e: e.time_range.start)
    names = [e.name for e in evs]
    print(f"# TrainStep, 2 ranks over gloo on one GPU, 2048 rays per rank: GPU activities of ONE step, in start order")
    for k, e in enumerate(evs):
        print(f"{k:3d} {e.time_range.start - evs[0].time_range.start:9.1f} us  {e.name[:110]}")
    print("# between a network's last backward kernel (wgrad_reduce_kernel) and its adam_kernel:")
    reduces = [k for k, n in enumerate(names) if "wgrad_reduce_kernel" in n]
    adams = [k for k, n in enumerate(names) if "adam_kernel" in n]
    for r in reduces:
        nxt = [a for a in adams if a > r]
        if nxt:
            between = names[r + 1:nxt[0]]
            launches = [n for n in between if "Memcpy" not in n and "memcpy" not in n and "copyBuffer" not in n]
            print(f"#   after reduce @{r}: {len(between)} activities before adam @{nxt[0]}, of which kernels (not gloo's staging copies): "
                  f"{len(launches)}  {[n[:60] for n in launches]}")
torch.distributed.barrier()
torch.distributed.destroy_process_group()
