#!/usr/bin/env python
"""Headline benchmark: training rays/s of the PL-NeRF hot path on MI355X.

A "step" is one optimisation step of run_plnerf.py:1259-1316 on synthetic 800x800 Blender-style views:
choose N_rand random pixels of a view and build their rays (device-side, inside the timed step) -> render (coarse
64 + fine 64+128 samples, piecewise-linear quadrature, exact PL importance sampling) -> mse(rgb)+mse(rgb0) ->
backward through both MLPs -> per-network gradient all-reduce across ranks (overlapped with the backward) -> Adam on
both networks.  N_rand = 4096 rays PER GPU (BASELINE.json configs[1]; configs[2] is the same per-GPU load at 8 GPUs
=> weak scaling).

    python bench.py --gpus N --steps K --warmup W          (N > 1 without torchrun's environment: this script spawns
                                                            its own N ranks, one per GPU, and rank 0 prints the line)
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

Other BASELINE configurations run with --workload {blender_128_64, llff_ndc, depth_128_64}; the default
(blender_64_128) is the one the metric is quoted on.

Rank 0 prints ONE JSON line.  `roofline` is measured live with HIP events around the dominant kernel (the fused
fine-network MLP forward) on the launch stream and priced per SURVEY.md section 8d: algorithmic FLOP / launch time /
dense 16-bit MFMA peak; `strict_fp32` re-times the step with the exact-fp32 kernels; `cpu_baseline` times the CPU
oracle (a port of the reference's PyTorch path) on the host cores on a bounded sample.  The oracle is never on the
measured GPU path.
"""
import argparse
import json
import os
import socket
import statistics
import subprocess
import sys
import tempfile
import time
from argparse import Namespace

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

FWD_FLOP_PER_ROW = 1186816      # 2 x 593,408 MAC per network evaluation -- the ALGORITHMIC work of the reference's layers
TRAIN_FLOP_PER_ROW = 3489024    # forward + wgrad + dgrad
# What the 16-bit modes' kernels EXECUTE per row since round 6: feature_linear is composed into the view layer (no activation
# between them, run_nerf_helpers.py:115-121), so the 256 x 256 feature GEMM (65,536 MACs) is not evaluated; exact fp32 keeps
# the reference's layers.  `roofline.frac` prices the algorithmic figure (SURVEY's contract), `frac_executed` this one.
FWD_FLOP_PER_ROW_EXECUTED = {"fp32": 1186816, "h16": 1186816 - 2 * 65536}
TRAFFIC_FILE = "r06_traffic.json"   # refreshed per round by tools/pmc_traffic.sh
PSNR_FILE = "r06_psnr_plain_modes.jsonl"      # tools/psnr_plain_modes.py on the round's kernels (the throughput mode's dB figure)
HBM_PEAK_GBS = 8000.0           # MI355X_MICROARCH.md: 8 TB/s HBM3E
# training forward, algorithmic bytes per row: saved state written + xyz read (12) + raw written (16)
FWD_TRAIN_BYTES_PER_ROW = {"fp32": 2596 * 4 + 28, "h16": 2272 * 2 + 272 + 28}   # fp32 planes | half planes + relu masks
PEAK_TFLOPS = {"fp32": 157.3, "bf16x3": 2500.0, "bf16": 2500.0, "f16x3": 2500.0, "f16": 2500.0}   # MI355X_MICROARCH.md, dense
MFMA_PER_PRODUCT = {"fp32": 1, "bf16x3": 3, "bf16": 1, "f16x3": 3, "f16": 1}
DTYPE = {"fp32": "f32", "bf16x3": "bf16x3 (3-term bf16 split, f32 accumulate)", "bf16": "bf16 (f32 accumulate)",
         "f16x3": "f16x3 (3-term f16 split, f32 accumulate)", "f16": "f16 (f32 accumulate)"}
FWD_KERNEL = {"fp32": "mlp_fwd_f32_kernel<2,true>", "f16x3": "mlp_fwd_rr_kernel<2,true>", "f16": "mlp_fwd_pp_kernel<1,true>",
              "bf16x3": "mlp_fwd_rr_kernel<2,true> [bf16 elements]", "bf16": "mlp_fwd_pp_kernel<1,true>"}
# (PLNERF_FWD_KERNEL=pp in the environment puts the split modes back on mlp_fwd_pp_kernel<2,true>; the line then names that one)
if os.environ.get("PLNERF_FWD_KERNEL") == "pp":
    FWD_KERNEL["f16x3"] = FWD_KERNEL["bf16x3"] = "mlp_fwd_pp_kernel<2,true>"
elif os.environ.get("PLNERF_FWD_KERNEL") == "rr":
    FWD_KERNEL["f16"] = "mlp_fwd_rr_kernel<1,true>"

# (N_samples, N_importance, description); every workload is 4096 rays per GPU, mode = linear / midpoint
WORKLOADS = {
    "blender_64_128": (64, 128, "BASELINE configs[1]: 800x800 Blender-style views, white background, perturb=1"),
    "blender_128_64": (128, 64, "configs/blender_linear.txt's own sampling (N_samples 128 / N_importance 64), otherwise "
                                "as configs[1]"),
    "llff_ndc": (64, 128, "BASELINE configs[3]: 378x504 forward-facing views, NDC rays (near 0, far 1), "
                          "raw_noise_std=1, no white background"),
    "depth_128_64": (128, 64, "BASELINE configs[4] per GPU: depth-supervised variant (57|3-channel network, softplus "
                              "density, space-carving loss through pred_hyp, clipped single Adam), N_samples 128 / "
                              "N_importance 64"),
}


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--rays", type=int, default=4096, help="N_rand per GPU")
    ap.add_argument("--workload", default="blender_64_128", choices=sorted(WORKLOADS))
    ap.add_argument("--n-samples", type=int, default=None, help="override the workload's N_samples")
    ap.add_argument("--n-importance", type=int, default=None, help="override the workload's N_importance")
    ap.add_argument("--precision", default="f16x3", choices=["fp32", "bf16x3", "bf16", "f16x3", "f16"],
                    help="MLP arithmetic.  f16x3 (default): every forward product as a 3-term IEEE-half split on the "
                         "16-bit MFMA pipe (holds the 1e-5 parity bound with ~7x margin); backward on IEEE-half planes "
                         "under one power-of-two scale per launch, single half MFMAs (the arithmetic of a loss-scaled "
                         "fp16 training step; gradient tolerance stated in tests/test_gpu_modes.py).  bf16x3: the same "
                         "with bf16 forward elements.  fp32: exact fp32 MFMA, forward and backward.  bf16 / f16: plain "
                         "16-bit operands (throughput only, outside the 1e-5 contract)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-strict-fp32", action="store_true", help="skip the exact-fp32 leg")
    ap.add_argument("--cpu-rays", type=int, default=1024)
    ap.add_argument("--cpu-threads", type=int, default=16)
    ap.add_argument("--cpu-no-grid", action="store_true", help="cpu_baseline: only the workload's own cell")
    ap.add_argument("--views", type=int, default=4, help="synthetic views resident on the device")
    ap.add_argument("--no-extra-legs", action="store_true",
                    help="skip the bounded extra legs of the default single-GPU run (north_star's 65,536 x 192 MLP benchmark "
                         "in every precision, the other BASELINE workloads x 10 steps, one 800x800 frame, the 200-step "
                         "PSNR-vs-fp32 run); they add ~25 s to the run and nothing to the timed region")
    ap.add_argument("--psnr-steps", type=int, default=200, help="steps of the psnr_vs_ref leg (each precision)")
    ap.add_argument("--force-dist", action="store_true",
                    help="initialise RCCL and run the gradient all-reduce even with one rank (path check)")
    ap.add_argument("--stub-cpu", action="store_true",
                    help="launcher self-test (tests/test_host_cpu.py): ranks on the CPU over gloo, the step is a stand-in "
                         "with NeRF-sized flat gradients through dp.GradientBucket; the line it prints is NOT a "
                         "measurement of the hot path and says so")
    ap.add_argument("--stub-fail-rank", type=int, default=-1,
                    help="launcher self-test: this rank raises before the first step (the launcher must stop the others "
                         "and exit non-zero)")
    ap.add_argument("--launch-timeout", type=float, default=1800.0, help="seconds the self-spawned ranks may run")
    return ap.parse_args(argv)


def make_args(a, ckpt_dir, precision):
    llff = a.workload == "llff_ndc"
    return Namespace(multires=10, i_embed=0, use_viewdirs=True, multires_views=4, N_importance=a.n_importance,
                     N_samples=a.n_samples, netdepth=8, netwidth=256, netdepth_fine=8, netwidth_fine=256,
                     netchunk=65536, lrate=5e-4, coarse_lrate=5e-4, ft_path=None, ckpt_dir=ckpt_dir, expname="exp",
                     no_reload=True, perturb=1.0, white_bkgd=not llff, raw_noise_std=1.0 if llff else 0.0, mode="linear",
                     color_mode="midpoint", dataset="llff" if llff else "blender", no_ndc=False, lindisp=False,
                     precision=precision, lrate_decay=500, constant_init=0, chunk=32768, N_rand=a.rays)


def depth_args(a, precision):
    return Namespace(multires=9, i_embed=0, use_viewdirs=True, multires_views=0, input_ch_cam=0,
                     N_importance=a.n_importance, N_samples=a.n_samples, netdepth=8, netwidth=256, netdepth_fine=8,
                     netwidth_fine=256, netchunk=65536, lrate=5e-4, perturb=1.0, white_bkgd=True, raw_noise_std=0.0,
                     mode="linear", color_mode="midpoint", lindisp=False, no_reload=True, space_carving_weight=0.007,
                     warm_start_nerf=0, is_joint=False, norm_p=2, space_carving_threshold=0.0, precision=precision,
                     bb_center=0.0, bb_scale=1.0)


def cpu_info():
    """Host description for the cpu_baseline object: model, logical CPUs, physical cores (sockets x cores per socket)."""
    model, sockets, per_socket = None, set(), None
    try:
        for line in open("/proc/cpuinfo"):
            key, _, val = line.partition(":")
            key = key.strip().lower()
            if key == "model name" and model is None:
                model = val.strip()
            elif key == "physical id":
                sockets.add(val.strip())
            elif key == "cpu cores" and per_socket is None:
                per_socket = int(val)
    except (OSError, ValueError):
        pass
    phys = (len(sockets) or 1) * per_socket if per_socket else None
    return {"cpu_count": os.cpu_count(), "physical_cores": phys, "sockets": len(sockets) or None, "cpu_model": model}


def cpu_cell(orc, n, ns, ni, reps):
    """One cell of BASELINE.md section 4's grid: the oracle's full training step, 1 warm-up + `reps` timed steps."""
    batch, target = orc.synthetic_blender_rays(n, seed=0)
    sd_c, sd_f = orc.closed_form_state_dict(0), orc.closed_form_state_dict(1)
    kw = dict(N_samples=ns, N_importance=ni, mode="linear", color_mode="midpoint", perturb=1.0, white_bkgd=True,
              raw_noise_std=0.0)
    state = {}
    orc.train_step(sd_c, sd_f, batch, target, kw, adam_state=state)
    t0 = time.perf_counter()
    for _ in range(reps):
        orc.train_step(sd_c, sd_f, batch, target, kw, adam_state=state)
    return (time.perf_counter() - t0) / reps


def cpu_baseline(a):
    """The oracle's training step on the host cores, bounded sample: the workload's own sampling at `--cpu-rays` rays
    is `value`; the other cells of BASELINE.md section 4's grid (R in {256, 1024} x {(64,128), (128,64)}) ride along in
    `grid`, 1 warm-up + 2 steps each (about 35 s of host time in total)."""
    from oracle import plnerf_oracle as orc
    # 16 threads is the fastest setting on the GPU box's 256-thread host for this workload
    # (profiles/r01_cpu_oracle_thread_sweep.txt: 8 -> 314, 16 -> 339, 32 -> 289, 64 -> 161,
    # 128 -> 71 rays/s; more threads only add oversubscription on these small GEMMs)
    threads = min(a.cpu_threads, os.cpu_count() or 1)
    torch.set_num_threads(threads)
    reps = 2
    cells = [(a.cpu_rays, a.n_samples, a.n_importance)]
    if not a.cpu_no_grid:
        for n in (256, 1024):
            for ns, ni in ((64, 128), (128, 64)):
                if (n, ns, ni) not in cells:
                    cells.append((n, ns, ni))
    grid = []
    for n, ns, ni in cells:
        dt = cpu_cell(orc, n, ns, ni, reps)
        grid.append({"rays": n, "N_samples": ns, "N_importance": ni, "s_per_step": dt, "rays_per_s": n / dt})
    head = grid[0]
    used = torch.get_num_threads()
    out = {"value": head["rays_per_s"], "unit": "rays/s", "cores": used, "threads": used, "kind": "port",
           "sample": f"{head['rays']} rays x ({a.n_samples} coarse + {a.n_samples + a.n_importance} fine) samples, full train step "
                     f"(fwd+bwd+2xAdam), fp32 PyTorch CPU oracle, 1 warm-up + mean of {reps} steps, "
                     f"{head['s_per_step']:.2f} s/step; `cores` = the {used} torch threads actually used, the fastest "
                     f"setting on this host (profiles/r01_cpu_oracle_thread_sweep.txt); the host has `physical_cores`",
           "grid": grid}
    out.update(cpu_info())
    return out


class Scene:
    """Synthetic views resident in HBM: poses on the NeRF-synthetic sphere (or near-identity forward-facing poses)
    and their target images (the analytic sphere scene of tools/scene.py).  What the reference's loader would have put on
    the device."""

    def __init__(self, P, workload, n_views, dev):
        gen = torch.Generator().manual_seed(1)
        if workload == "llff_ndc":
            self.H, self.W, f = 378, 504, 407.0
            self.near, self.far = 0.0, 1.0
            poses = []
            for i in range(n_views):
                c2w = torch.eye(4)[:3, :4].clone()
                c2w[:, 3] = torch.tensor([0.05 * i - 0.1, 0.02 * i, 0.1])
                poses.append(c2w)
        else:
            self.H, self.W, f = 800, 800, 1111.111
            self.near, self.far = 2.0, 6.0
            poses = [P.rays.pose_spherical(-180.0 + 360.0 * i / max(n_views, 1), -30.0, 4.0)[:3, :4]
                     for i in range(n_views)]
        self.K = [[f, 0, self.W / 2], [0, f, self.H / 2], [0, 0, 1]]
        self.poses = poses
        if workload == "llff_ndc":      # (no analytic forward-facing scene: smooth colour ramps)
            yy, xx = torch.meshgrid(torch.linspace(0, 1, self.H), torch.linspace(0, 1, self.W), indexing="ij")
            self.images = [torch.stack([xx, yy, 0.5 * (xx + yy)], -1).roll(37 * i, 1).contiguous().to(dev)
                           for i in range(n_views)]
        else:
            # targets a network can fit: the analytic sphere scene of tools/scene.py seen from each pose (round 5; until
            # round 4 uniform noise, whose loss never fell -- same step time, profiles/r04_psnr_vs_fp32_2000steps.jsonl)
            from tools.scene import analytic_image
            self.images = [analytic_image(P, self.H, self.W, self.K, c2w, dev) for c2w in poses]
        self.hyp = None
        if workload == "depth_128_64":     # three depth hypotheses per pixel (target_h of the space-carving loss)
            self.hyp = [(2.0 + 4.0 * torch.rand(3, self.H, self.W, generator=gen)).to(dev) for _ in range(n_views)]


def build_step(P, a, precision, scene, dev, rank, world, force_dist):
    """Returns (step(i) -> loss, nets) for the workload in the given arithmetic."""
    from plnerf_amd import dp
    ck = tempfile.mkdtemp()
    os.makedirs(os.path.join(ck, "exp"))
    torch.manual_seed(0)
    if a.workload == "depth_128_64":
        from plnerf_amd import depth as Dp
        args = depth_args(a, precision)
        kw, _, _, grad_vars, opt = Dp.create_nerf(args, device=dev)
        nets = [kw["network_fn"], kw["network_fine"]]
        ts = Dp.DepthTrainStep(args, kw, opt, grad_vars, distributed=world > 1, seed=0)      # (counter-based draws: sharding-invariant)
        if force_dist and world == 1:
            ts.bucket = dp.GradientBucket(nets)

        def step(i):
            v = i % len(scene.poses)
            cols, target, pix = P.select_view_rays(scene.H, scene.W, scene.K, scene.poses[v], scene.images[v], a.rays,
                                                   scene.near, scene.far, seed=0, step=i, ray_id0=rank * a.rays,
                                                   want_pixels=True)
            target_h = scene.hyp[v][:, pix[:, 0].long(), pix[:, 1].long()].unsqueeze(-1)
            loss, _, _, _ = ts(cols, target, target_h)
            return loss
        step.trainer = ts
        return step, nets
    args = make_args(a, ck, precision)
    _stdout = sys.stdout
    sys.stdout = open(os.devnull, "w")
    try:
        kw, _, _, _, opt, opt_c = P.create_nerf(args, device=dev)
    finally:
        sys.stdout = _stdout
    nets = [kw["network_fn"], kw["network_fine"]]
    ts = P.TrainStep(args, kw, opt, opt_c, distributed=world > 1, seed=0)
    if force_dist and world == 1:
        ts.bucket = dp.GradientBucket(nets)
    torch.manual_seed(1000 + rank)         # torch.randn density noise (llff): decorrelate the ranks

    def step(i):
        v = i % len(scene.poses)
        loss, _ = ts.step_view(scene.H, scene.W, scene.K, scene.poses[v], scene.images[v], near=scene.near,
                               far=scene.far, n_rand=a.rays)
        return loss
    step.trainer = ts
    return step, nets


def free_port():
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def launch_ranks(a, argv):
    """`bench.py --gpus N` without torchrun's environment: spawn the N ranks here (one process per GPU, LOCAL_RANK ->
    device, a free rendezvous port on 127.0.0.1), let rank 0 print the one JSON line on this process's stdout, and
    exit non-zero if any rank fails (the others are then terminated by PID)."""
    n = a.gpus
    if not a.stub_cpu:
        have = torch.cuda.device_count()
        if have < n:
            print(f"bench.py: --gpus {n} needs {n} visible devices, this host has {have} "
                  f"(nothing was launched)", file=sys.stderr)
            return 2
    port = free_port()
    procs = []
    threads = rank_threads(n)
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), PLNERF_BENCH_SELF_LAUNCHED="1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # a rank is ONE Python launch loop (~1.2 ms of host time per 5.6 ms step): without a bound every rank's torch / OpenMP
        # runtime spawns a thread per logical CPU (torchrun sets OMP_NUM_THREADS=1 for the same reason)
        env.setdefault("OMP_NUM_THREADS", str(threads))
        env.setdefault("MKL_NUM_THREADS", str(threads))
        # rank 0 inherits stdout (the JSON line); whatever the other ranks print goes to stderr
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=None if r == 0 else sys.stderr))
    deadline = time.monotonic() + a.launch_timeout
    rc = 0
    live = set(range(n))
    while live:
        for r in sorted(live):
            code = procs[r].poll()
            if code is None:
                continue
            live.discard(r)
            if code != 0 and rc == 0:
                rc = code if code > 0 else 1
                print(f"bench.py: rank {r} exited with {code}; stopping the other ranks", file=sys.stderr)
                for q in live:
                    procs[q].terminate()
        if live and time.monotonic() > deadline:
            print(f"bench.py: ranks {sorted(live)} still running after {a.launch_timeout:.0f} s; stopping them",
                  file=sys.stderr)
            for q in live:
                procs[q].kill()
            rc = rc or 124
            deadline = float("inf")
        if live:
            time.sleep(0.05)
    return rc


def rank_threads(world):
    """Host threads a rank may use: the logical CPUs shared out over the ranks, at most 8 (the step needs one)."""
    return max(1, min(8, (os.cpu_count() or 1) // max(world, 1)))


def gpu_numa_node(local_rank):
    """NUMA node of this rank's GPU from the KFD topology (the local_rank-th node that has SIMDs), or None."""
    base = "/sys/class/kfd/kfd/topology/nodes"
    try:
        gpus = []
        for node in sorted(os.listdir(base), key=int):
            try:      # (a container that is handed some of the node's GPUs may not read the others' entries: they are not its ranks')
                props = dict(l.split()[:2] for l in open(os.path.join(base, node, "properties")) if len(l.split()) >= 2)
            except OSError:
                continue
            if int(props.get("simd_count", "0")) > 0:
                gpus.append(props)
        domain, loc = int(gpus[local_rank]["domain"]), int(gpus[local_rank]["location_id"])
        bdf = f"{domain:04x}:{(loc >> 8) & 0xff:02x}:{(loc >> 3) & 0x1f:02x}.{loc & 7:x}"
        node = int(open(f"/sys/bus/pci/devices/{bdf}/numa_node").read())
        return node if node >= 0 else None
    except Exception:
        return None


def pin_rank(local_rank, world):
    """Bind this rank to the cores of its GPU's NUMA node (a 2-socket host: the launch loop and RCCL's proxy thread next to the
    device they feed) and bound its intra-op threads.  Silent when the topology is unreadable.  Returns what was done."""
    threads = int(os.environ.get("OMP_NUM_THREADS") or rank_threads(world))
    torch.set_num_threads(threads)
    info = {"omp_num_threads": threads, "numa_node": None, "cpus": None}
    if world < 2 or not hasattr(os, "sched_setaffinity"):
        return info
    node = gpu_numa_node(local_rank)
    if node is None:
        return info
    try:
        cpus = set()
        for part in open(f"/sys/devices/system/node/node{node}/cpulist").read().strip().split(","):
            lo, _, hi = part.partition("-")
            cpus.update(range(int(lo), int(hi or lo) + 1))
        cpus &= os.sched_getaffinity(0)
        if cpus:
            os.sched_setaffinity(0, cpus)
            info.update(numa_node=node, cpus=len(cpus))
    except Exception:
        pass
    return info


_PROBES = None


def mfma_sustained_tflops():
    """tools/probes/libplnerf_probes.so (built by __graft_entry__.build(); a measurement aid, not the product library): the
    rate a pure v_mfma_f32_32x32x16_f16 stream sustains on toggling operands, measured in this process (~0.2 s)."""
    global _PROBES
    import ctypes
    path = os.path.join(ROOT, "tools", "probes", "libplnerf_probes.so")
    if not os.path.exists(path):
        return None
    if _PROBES is None:
        _PROBES = ctypes.CDLL(path)
        _PROBES.plnerf_probe_mfma_tflops.restype = ctypes.c_double
        _PROBES.plnerf_probe_mfma_tflops.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    tf = _PROBES.plnerf_probe_mfma_tflops(40000, 1, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    return tf if tf > 0 else None


def build_stub_step(rank, world):
    """Launcher self-test (--stub-cpu): two NeRF-sized flat gradient buffers through dp.GradientBucket on CPU / gloo.
    Not the hot path; exists so that the spawn + rendezvous + collective + report plumbing runs without GPUs."""
    from plnerf_amd import dp
    torch.manual_seed(7 + rank)
    nets = [torch.nn.Linear(64, 32), torch.nn.Linear(32, 4)]
    dp.broadcast_parameters(nets)
    bucket = dp.GradientBucket(nets)
    x = torch.randn(16, 64)

    def step(i):
        for n in nets:
            n.zero_grad()
        loss = (nets[1](torch.relu(nets[0](x))) ** 2).mean()
        loss.backward()
        t0 = time.perf_counter()
        bucket.allreduce_mean()
        step.allreduce_s.append(time.perf_counter() - t0)
        return loss
    step.allreduce_s = []
    return step, nets


# ---------------------------------------------------------------------------------------------------------------------
# Bounded extra legs of the default single-GPU run: every number the documents quote, on the one
# line the driver records.  None of them is inside the timed region of `value`.
# ---------------------------------------------------------------------------------------------------------------------
def leg_mlp_only(P, dev, rays=65536, samples=192, iters=3):
    """north_star's MLP benchmark: the fused PE + MLP forward (inference) at 65,536 x 192 rows, every precision, each
    with its roofline on the algorithmic FLOP (14.93 TFLOP per launch)."""
    torch.manual_seed(0)
    pts = (torch.rand(rays, samples, 3, device=dev) * 2 - 1) * 3
    vd = torch.nn.functional.normalize(torch.randn(rays, 3, device=dev), dim=-1)
    rows = rays * samples
    out = {"rows": rows, "what": "fused PE+MLP forward, inference (no saved state), one launch", "flop_per_row": FWD_FLOP_PER_ROW}
    for prec in ("f16x3", "bf16x3", "f16", "bf16", "fp32"):
        net = P.NeRF(D=8, W=256, input_ch=63, input_ch_views=27, output_ch=5, skips=[4], use_viewdirs=True,
                     precision=prec).to(dev)
        with torch.no_grad():
            net.query(pts, vd)
            torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(iters):
                net.query(pts, vd)
            e.record()
            torch.cuda.synchronize()
        ms = s.elapsed_time(e) / iters
        tf = rows * FWD_FLOP_PER_ROW / (ms * 1e-3) / 1e12
        ex = FWD_FLOP_PER_ROW_EXECUTED["fp32" if prec == "fp32" else "h16"] / FWD_FLOP_PER_ROW
        out[prec] = {"ms": ms, "roofline": {"bound": "mfma", "achieved": tf, "peak": PEAK_TFLOPS[prec], "unit": "TFLOP/s",
                                           "frac": tf / PEAK_TFLOPS[prec], "frac_executed": ex * tf / PEAK_TFLOPS[prec]},
                     # (bf16x3 holds 8.2e-6 on the network output but reaches 2.9e-5 on render_rays' maps: a 3e-5 mode)
                     "holds_1e-5_contract": prec in ("f16x3", "fp32"),
                     "render_rays_bound": {"f16x3": 1e-5, "fp32": 1e-5, "bf16x3": 3e-5}.get(prec)}
        del net
    del pts, vd
    torch.cuda.empty_cache()
    return out


def leg_frame(P, dev, precision="f16x3"):
    """One 800 x 800 view through render(c2w=...) under no_grad: 640,000 rays in 32,768-ray
    chunks, 64 + 192 samples; priced on the frame's algorithmic MLP FLOP."""
    from tools.scene import blender_intrinsics, nerf_args
    ck = tempfile.mkdtemp()
    os.makedirs(os.path.join(ck, "exp"))
    torch.manual_seed(0)
    so, sys.stdout = sys.stdout, open(os.devnull, "w")
    try:
        _, kw_test, _, _, _, _ = P.create_nerf(nerf_args(precision, ck), device=dev)
    finally:
        sys.stdout = so
    H = W = 800
    K = blender_intrinsics(H, W)
    poses = [P.rays.pose_spherical(th, -30.0, 4.0)[:3, :4].to(dev) for th in (0.0, 90.0)]
    with torch.no_grad():
        P.render(H, W, K, chunk=32768, c2w=poses[0], near=2.0, far=6.0, **kw_test)
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        rgb, _, _, _ = P.render(H, W, K, chunk=32768, c2w=poses[1], near=2.0, far=6.0, **kw_test)
        e.record()
        torch.cuda.synchronize()
    ms = s.elapsed_time(e)
    tf = H * W * (64 + 192) * FWD_FLOP_PER_ROW / (ms * 1e-3) / 1e12
    out = {"precision": precision, "ms_per_frame": ms, "rays_per_s": H * W / (ms * 1e-3), "chunk": 32768,
           "roofline": {"bound": "mfma", "achieved": tf, "peak": PEAK_TFLOPS[precision], "unit": "TFLOP/s",
                        "frac": tf / PEAK_TFLOPS[precision], "note": "whole frame: algorithmic MLP FLOP / frame time"},
           "finite": bool(torch.isfinite(rgb).all())}
    del kw_test, rgb
    torch.cuda.empty_cache()
    return out


def peak_of(precision):
    return PEAK_TFLOPS[precision]


def stats_ms(v):
    return {"min": min(v), "median": statistics.median(v), "max": max(v)} if v else None


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    a = parse(argv)
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(a, argv))
    ns, ni, desc = WORKLOADS[a.workload]
    a.n_samples = a.n_samples if a.n_samples is not None else ns
    a.n_importance = a.n_importance if a.n_importance is not None else ni
    import plnerf_amd as P
    from plnerf_amd import dp, functional as Fn
    cpu = a.stub_cpu
    rank, world, local = dp.init_from_env(backend="gloo" if cpu else None, force=a.force_dist)
    if world != a.gpus:
        print(f"bench.py: --gpus {a.gpus} but the environment says WORLD_SIZE={world}", file=sys.stderr)
        sys.exit(2)
    if not cpu:
        if torch.cuda.device_count() <= local:
            print(f"bench.py: rank {rank} wants device {local}, this host has {torch.cuda.device_count()}",
                  file=sys.stderr)
            sys.exit(2)
        torch.cuda.set_device(local)
    dev = torch.device("cpu") if cpu else torch.device("cuda", local)
    dist_on = world > 1 or a.force_dist
    pinned = pin_rank(local, world)

    if cpu:
        if rank == a.stub_fail_rank:
            raise RuntimeError(f"--stub-fail-rank {rank}: simulated rank failure")
        step, nets = build_stub_step(rank, world)
        scene = None
    else:
        scene = Scene(P, a.workload, a.views, dev)
        step, nets = build_step(P, a, a.precision, scene, dev, rank, world, a.force_dist)
    if a.force_dist and world == 1 and not cpu:     # exercise the RCCL path on one GPU
        _finish = dp.GradientBucket.finish
        dp.GradientBucket.finish = lambda self, modules=None, defer_scale=False, group=None, force=False: \
            _finish(self, modules, defer_scale, group, True)
    R = a.rays
    rows_fine = R * (a.n_samples + a.n_importance)

    def sync():
        if dist_on:
            torch.distributed.barrier(**({} if cpu else {"device_ids": [local]}))
        if not cpu:
            torch.cuda.synchronize()

    def timed(step_fn, warmup, steps, timer=None):
        """W untimed steps, then exactly K steps between barrier + synchronize on both sides; MAX over ranks.
        Also returns every rank's own time and, on the GPU, the per-step times from one HIP event per step boundary
        (recorded on the launch stream: no host synchronisation inside the timed region)."""
        for i in range(warmup):
            step_fn(i)
        Fn.KERNEL_TIMER = timer
        marks = []
        sync()
        t0 = time.perf_counter()
        for i in range(warmup, warmup + steps):
            if not cpu:
                ev = torch.cuda.Event(enable_timing=True)
                ev.record()
                marks.append(ev)
            loss = step_fn(i)
        if not cpu:
            ev = torch.cuda.Event(enable_timing=True)
            ev.record()
            marks.append(ev)
        timed.host_s = time.perf_counter() - t0      # the host has ENQUEUED the K steps (tools/host_overhead.py's measure)
        sync()
        dt = time.perf_counter() - t0
        Fn.KERNEL_TIMER = None
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        per_rank = [dt]
        timed.host_per_rank = [timed.host_s]
        if dist_on:
            every = [torch.zeros(2, device=dev, dtype=torch.float64) for _ in range(world)]
            torch.distributed.all_gather(every, torch.tensor([dt, timed.host_s], device=dev, dtype=torch.float64))
            per_rank = [float(x[0].item()) for x in every]
            timed.host_per_rank = [float(x[1].item()) for x in every]
            torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        per_step = [marks[k].elapsed_time(marks[k + 1]) for k in range(len(marks) - 1)]
        return float(t.item()), float(loss.detach()), per_rank, per_step

    timer = None if cpu else Fn.KernelTimer()
    dt, final_loss, per_rank_s, per_step_ms = timed(step, a.warmup, a.steps, timer)
    host_per_rank_s = list(timed.host_per_rank)
    merged_steps = getattr(getattr(step, "trainer", None), "merged_steps", None)

    strict = None
    if a.precision != "fp32" and not a.no_strict_fp32 and world == 1 and not cpu:
        # the strictly reference-equal arithmetic (exact fp32 MFMA forward and backward), same workload, short leg
        step32, nets32 = build_step(P, a, "fp32", scene, dev, rank, world, False)
        s_steps = max(3, min(a.steps, 5))
        dt32, _, _, _ = timed(step32, 2, s_steps)
        strict = {"precision": "fp32", "ms_per_step": 1e3 * dt32 / s_steps, "rays_per_s": R * s_steps / dt32,
                  "steps": s_steps, "warmup": 2}
        del step32, nets32
        torch.cuda.empty_cache()

    extra = {}
    if world == 1 and not cpu and not a.no_extra_legs and a.workload == "blender_64_128" and not a.force_dist:
        # -- the other BASELINE workloads, 10 steps each (configs[3], configs[4] per GPU, the config file's own 128 + 64)
        import copy
        wl = {}
        for w in ("blender_128_64", "llff_ndc", "depth_128_64"):
            b = copy.copy(a)
            b.workload, (b.n_samples, b.n_importance) = w, WORKLOADS[w][:2]
            sc = Scene(P, w, a.views, dev)
            st, nets_w = build_step(P, b, a.precision, sc, dev, rank, world, False)
            tm = Fn.KernelTimer()
            dtw, _, _, psw = timed(st, 3, 10, tm)
            rows_w = R * (b.n_samples + b.n_importance)
            fwd_w = tm.mean_ms(f"mlp_fwd[{rows_w}]")
            tfw = rows_w * FWD_FLOP_PER_ROW / (fwd_w * 1e-3) / 1e12 if fwd_w else None
            wl[w] = {"ms_per_step": 1e3 * dtw / 10, "rays_per_s": R * 10 / dtw, "steps": 10, "warmup": 3,
                     "step_ms": stats_ms(psw), "fine_fwd_launch_ms": fwd_w, "fine_fwd_rows": rows_w,
                     "fine_fwd_frac_of_mfma_peak": (tfw / peak_of(a.precision)) if tfw else None,
                     "what": WORKLOADS[w][2]}
            del st, nets_w, sc, tm
            torch.cuda.empty_cache()
        extra["workloads"] = wl
        # -- SURVEY H1's throughput mode (plain half operands, forward and backward: outside the 1e-5 contract, parity judged
        #    by PSNR): the same workload, 20 steps; the dB figure is the six-seed 2000-step comparison on file
        if a.precision == "f16x3":
            b = copy.copy(a)
            st, nets_w = build_step(P, b, "f16", scene, dev, rank, world, False)
            dtt, loss_t, _, pst = timed(st, 5, 20)
            tm = {"precision": "f16", "ms_per_step": 1e3 * dtt / 20, "rays_per_s": R * 20 / dtt, "steps": 20, "warmup": 5,
                  "step_ms": stats_ms(pst), "final_loss": loss_t, "holds_1e-5_contract": False,
                  "what": "plain IEEE-half operands on single MFMAs, forward and backward (raw error 7.5e-4 on the sharpened "
                          "network, tests/test_gpu_modes.py); secondary to the headline"}
            try:
                for line in open(os.path.join(ROOT, "profiles", PSNR_FILE)):
                    d = json.loads(line)
                    if d.get("summary"):
                        g = d["f16"]["gap_db_train"]
                        nf = d["noise_floor_db_train (fp32 twin - fp32, round 4's files)"]
                        tm.update({"psnr_gap_db": g["mean"], "psnr_gap_db_std": g["std"], "psnr_gap_db_seeds": len(g["values"]),
                                   "psnr_noise_floor_db": {"mean": nf["mean"], "std": nf["std"]},
                                   "psnr_gap_tolerance_db": 0.3,
                                   "psnr_source": f"profiles/{PSNR_FILE} (tools/psnr_plain_modes.py: 2000 steps x 4096 "
                                                  "rays x 6 seeds against exact fp32; offline, not measured in this run); bf16: "
                                                  f"{d['bf16']['gap_db_train']['mean']:+.2f} dB (std {d['bf16']['gap_db_train']['std']:.2f})"})
            except Exception:
                pass
            extra["throughput_mode"] = tm
            del st, nets_w
            torch.cuda.empty_cache()
        # -- north_star's MLP benchmark, every precision
        extra["mlp_only_65536x192"] = leg_mlp_only(P, dev)
        # -- one full frame through render(c2w=...)
        extra["frame_800x800"] = leg_frame(P, dev, a.precision)
        # -- "PSNR vs ref" (BASELINE.json's metric, second half): the benchmarked arithmetic against the exact-fp32 kernels
        if a.precision != "fp32" and a.psnr_steps > 0:
            from tools.scene import psnr_vs_ref
            extra["psnr_vs_ref"] = psnr_vs_ref(P, dev, a.psnr_steps, rays=R, precision=a.precision)
            torch.cuda.empty_cache()

    # the all-reduce as the launch stream sees it: HIP events on that stream either side of GradientBucket.allreduce_mean
    # (wait for the two collectives enqueued during the backward + the 1/world scaling) = the EXPOSED part of the exchange
    if cpu:
        exposed = [1e3 * x for x in step.allreduce_s[a.warmup:]]
    else:
        exposed = timer.all_ms("allreduce_exposed")
    ex_mean = torch.tensor([sum(exposed) / len(exposed) if exposed else 0.0], device=dev, dtype=torch.float64)
    if dist_on:
        torch.distributed.all_reduce(ex_mean, op=torch.distributed.ReduceOp.MAX)

    if rank == 0 and cpu:
        print(json.dumps({
            "metric": "launcher self-test (stub step on CPU / gloo; NOT a measurement of the hot path)",
            "value": world * a.steps / dt, "unit": "stub steps/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": 1e3 * dt / a.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "stub", "parallelism": f"dp{world}", "backend": torch.distributed.get_backend()
                       if torch.distributed.is_initialized() else None,
                       "rccl_world_size": torch.distributed.get_world_size() if torch.distributed.is_initialized() else 1,
                       "self_launched": bool(os.environ.get("PLNERF_BENCH_SELF_LAUNCHED"))},
            "ranks": {"ms_per_step": stats_ms([1e3 * x / a.steps for x in per_rank_s]),
                      "host_ms_per_step": stats_ms([1e3 * x / a.steps for x in host_per_rank_s]),
                      "omp_num_threads": pinned["omp_num_threads"], "numa_node_rank0": pinned["numa_node"],
                      "allreduce_exposed_ms_max_over_ranks": float(ex_mean.item())}}), flush=True)
    elif rank == 0:
        ms = 1e3 * dt / a.steps
        fwd_ms = timer.mean_ms(f"mlp_fwd[{rows_fine}]")
        bwd_ms = timer.mean_ms(f"mlp_bwd[{rows_fine}]")      # (autograd order: the fine network's backward on its own)
        rows_coarse = R * a.n_samples
        bwd_both_ms = timer.mean_ms(f"mlp_bwd[{rows_coarse}+{rows_fine}]")      # (merged: both networks, one launch sequence)
        peak = PEAK_TFLOPS[a.precision]
        traffic, traffic_source = None, None
        fwd_kernel = FWD_KERNEL[a.precision]
        if a.workload == "depth_128_64" and a.precision in ("f16", "bf16"):
            # the caller-embedded 57|3 input of the depth variant: the plain modes have no register-resident variant for it
            fwd_kernel = "mlp_fwd_pp_kernel<1,true>"
        try:   # measured offline with rocprofv3 --pmc (cannot be collected from inside this process): the JSON says so
            tj = json.load(open(os.path.join(ROOT, "profiles", TRAFFIC_FILE)))
            t = tj.get(a.precision)
            if t and t["rows_per_launch"] == rows_fine and t.get("kernel", fwd_kernel) == fwd_kernel:
                traffic = t["bytes"]
                traffic_source = f"profiles/{TRAFFIC_FILE}@{tj.get('commit', 'unknown')} (offline rocprofv3 --pmc " \
                                 f"FETCH_SIZE / WRITE_SIZE passes, tools/pmc_traffic.sh; not measured in this run)"
        except Exception:
            traffic = None
        ach = rows_fine * FWD_FLOP_PER_ROW / (fwd_ms * 1e-3) / 1e12 if fwd_ms else None
        h16 = a.precision != "fp32"
        flop_exec = FWD_FLOP_PER_ROW_EXECUTED["h16" if h16 else "fp32"]
        ach_exec = rows_fine * flop_exec / (fwd_ms * 1e-3) / 1e12 if fwd_ms else None
        issued = (ach_exec * MFMA_PER_PRODUCT[a.precision]) if ach_exec else None
        # (not under --no-extra-legs: the profiling runs of tools/ use that flag, and a 0.2 s MFMA stream would head their kernel stats)
        sustained = mfma_sustained_tflops() if (h16 and world == 1 and not a.no_extra_legs) else None
        took_merged = bool(merged_steps) and merged_steps >= a.warmup + a.steps
        bytes_per_row = FWD_TRAIN_BYTES_PER_ROW["h16" if h16 else "fp32"]
        ach_gbs = rows_fine * bytes_per_row / (fwd_ms * 1e-3) / 1e9 if fwd_ms else None
        out = {
            "metric": "training rays/sec (coarse+fine, 64+128 samples)" if a.workload == "blender_64_128" else
                      f"training rays/sec ({a.workload})",
            "value": R * world * a.steps / dt, "unit": "rays/s", "n_gpus": world, "steps": a.steps,
            "warmup": a.warmup, "ms_per_step": ms, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": DTYPE[a.precision], "data": "synthetic",
            "config": {"workload": f"{a.workload} -- {desc}; N_rand={R}/GPU, N_samples={a.n_samples}, "
                                   f"N_importance={a.n_importance}, mode=linear/midpoint; full step = device-side pixel "
                                   f"choice + ray generation + render + backward + "
                                   + ("one gradient all-reduce for both networks" if took_merged else
                                      "one gradient all-reduce per network") + " + Adam",
                       "global_rays": R * world, "precision": a.precision, "parallelism": f"dp{world}",
                       "rccl_world_size": torch.distributed.get_world_size() if torch.distributed.is_initialized() else 1,
                       "backend": torch.distributed.get_backend() if torch.distributed.is_initialized() else None,
                       "self_launched": bool(os.environ.get("PLNERF_BENCH_SELF_LAUNCHED")),
                       "merged_backward": took_merged, "merged_backward_steps": merged_steps,
                       "final_loss": final_loss},
            # the K timed steps one by one (rank 0's launch stream, one HIP event per step boundary), and the ranks'
            # own wall-clock times for the K steps: the spread is what diagnoses a slow rank / an exposed collective
            "step_ms": stats_ms(per_step_ms),
            "ranks": {"ms_per_step": stats_ms([1e3 * x / a.steps for x in per_rank_s]),
                      # wall time each rank's host needed to ENQUEUE a step (no synchronisation inside the timed region): a
                      # rank whose host loop is slower than its GPU shows here, an exposed collective shows below
                      "host_ms_per_step": stats_ms([1e3 * x / a.steps for x in host_per_rank_s]),
                      "omp_num_threads": pinned["omp_num_threads"], "numa_node_rank0": pinned["numa_node"],
                      "cpus_rank0": pinned["cpus"],
                      "allreduce_exposed_ms_max_over_ranks": float(ex_mean.item()) if dist_on else None,
                      "allreduce_exposed_ms_rank0": stats_ms(exposed)},
            # the MLP is priced against the MFMA roofline on its ALGORITHMIC work,
            # 1,186,816 FLOP per network evaluation (`frac`) -- the 3 MFMA issues per product of the split modes are a cost,
            # not work, and the feature GEMM the 16-bit modes no longer evaluate still counts as work done.  `frac_executed`
            # prices the FLOP the kernel really evaluates (flop_per_row_executed); `mfma_issued_tflops` = that x MFMAs per
            # product; `sustained_peak` = what a pure MFMA stream on toggling operands reaches on THIS box in THIS run
            # (tools/probes/mfma_sustained.hip), `frac_of_sustained` = issued / sustained.  north_star's ">= 50 % of the
            # bf16 MFMA roofline with render_rays within 1e-5" is MISSED: three MFMAs per product cap `frac` at 0.33 nominal.
            # The HBM view of the same launch (saved half planes written once) rides along.
            "roofline": {
                "bound": "mfma",
                "kernel": fwd_kernel + " (fine network, fused PE+12-layer MLP forward, saves backward state)",
                "achieved": ach, "peak": peak, "unit": "TFLOP/s", "frac": (ach / peak) if ach else None,
                "traffic": traffic, "traffic_source": traffic_source, "launch_ms": fwd_ms, "rows_per_launch": rows_fine,
                "flop_per_row": FWD_FLOP_PER_ROW,
                "flop_per_row_executed": flop_exec, "achieved_executed": ach_exec,
                "frac_executed": (ach_exec / peak) if ach_exec else None,
                "mfma_per_product": MFMA_PER_PRODUCT[a.precision],
                "mfma_issued_tflops": issued,
                "mfma_issue_frac": (issued / peak) if issued else None,
                "sustained_peak": sustained,
                "frac_of_sustained": (issued / sustained) if (issued and sustained) else None,
                "north_star_50pct_with_1e-5": "missed" if a.precision in ("f16x3", "bf16x3") else None,
                "hbm_view": {"bytes_per_row": bytes_per_row, "achieved_gbs": ach_gbs, "peak_gbs": HBM_PEAK_GBS,
                             "frac": (ach_gbs / HBM_PEAK_GBS) if ach_gbs else None},
                "mlp_bwd_launch_ms": bwd_ms,
                "mlp_bwd_both_networks_ms": bwd_both_ms,
                "train_mlp_tflops": (rows_fine * TRAIN_FLOP_PER_ROW / ((fwd_ms + bwd_ms) * 1e-3) / 1e12)
                if (fwd_ms and bwd_ms) else None,
                "step_tflops": (R * (2 * a.n_samples + a.n_importance) * TRAIN_FLOP_PER_ROW / (ms * 1e-3) / 1e12),
            },
        }
        if strict is not None:
            out["strict_fp32"] = strict
        out.update(extra)
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(a)
        # RCCL writes its version banner through C stdio, which is block-buffered on a pipe and would otherwise
        # come out at process exit, AFTER this line: flush it first so that the JSON is the last line of stdout
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        sys.stdout.flush()
        print(json.dumps(out), flush=True)
    if torch.distributed.is_initialized():
        if dist_on:
            torch.distributed.barrier(**({} if cpu else {"device_ids": [local]}))
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
