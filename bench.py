#!/usr/bin/env python
"""Headline benchmark: training rays/s of the PL-NeRF hot path on MI355X.

A "step" is one optimisation step of run_plnerf.py:1259-1316 on synthetic 800x800 Blender-style views:
choose N_rand random pixels of a view and build their rays (device-side, inside the timed step) -> render (coarse
64 + fine 64+128 samples, piecewise-linear quadrature, exact PL importance sampling) -> mse(rgb)+mse(rgb0) ->
backward through both MLPs -> per-network gradient all-reduce across ranks (overlapped with the backward) -> Adam on
both networks.  N_rand = 4096 rays PER GPU (BASELINE.json configs[1]; configs[2] is the same per-GPU load at 8 GPUs
=> weak scaling).

    python bench.py --gpus 1 --steps K --warmup W
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
import sys
import tempfile
import time
from argparse import Namespace

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

FWD_FLOP_PER_ROW = 1186816      # 2 x 593,408 MAC per network evaluation
TRAIN_FLOP_PER_ROW = 3489024    # forward + wgrad + dgrad
HBM_PEAK_GBS = 8000.0           # MI355X_MICROARCH.md: 8 TB/s HBM3E
# training forward, algorithmic bytes per row: saved state written + xyz read (12) + raw written (16)
FWD_TRAIN_BYTES_PER_ROW = {"fp32": 2596 * 4 + 28, "h16": 2528 * 2 + 272 + 28}   # fp32 planes | half planes + relu masks
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


def parse():
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
    ap.add_argument("--views", type=int, default=4, help="synthetic views resident on the device")
    ap.add_argument("--force-dist", action="store_true",
                    help="initialise RCCL and run the gradient all-reduce even with one rank (path check)")
    return ap.parse_args()


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
    model = None
    try:
        for line in open("/proc/cpuinfo"):
            if line.lower().startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return {"cpu_count": os.cpu_count(), "cpu_model": model}


def cpu_baseline(a):
    """The oracle's training step on the host cores, bounded sample (1 warm-up + 3 steps)."""
    from oracle import plnerf_oracle as orc
    n = a.cpu_rays
    # 16 threads is the fastest setting on the GPU box's 256-thread host for this workload
    # (profiles/r01_cpu_oracle_thread_sweep.txt: 8 -> 314, 16 -> 339, 32 -> 289, 64 -> 161,
    # 128 -> 71 rays/s; more threads only add oversubscription on these small GEMMs)
    threads = min(a.cpu_threads, os.cpu_count() or 1)
    torch.set_num_threads(threads)
    batch, target = orc.synthetic_blender_rays(n, seed=0)
    sd_c, sd_f = orc.closed_form_state_dict(0), orc.closed_form_state_dict(1)
    kw = dict(N_samples=a.n_samples, N_importance=a.n_importance, mode="linear", color_mode="midpoint",
              perturb=1.0, white_bkgd=True, raw_noise_std=0.0)
    state = {}
    orc.train_step(sd_c, sd_f, batch, target, kw, adam_state=state)
    t0 = time.perf_counter()
    reps = 3
    for _ in range(reps):
        orc.train_step(sd_c, sd_f, batch, target, kw, adam_state=state)
    dt = (time.perf_counter() - t0) / reps
    out = {"value": n / dt, "unit": "rays/s", "cores": torch.get_num_threads(), "threads": torch.get_num_threads(),
           "kind": "port",
           "sample": f"{n} rays x ({a.n_samples}+{a.n_samples + a.n_importance}) samples, full train step "
                     f"(fwd+bwd+2xAdam), fp32 PyTorch CPU oracle, 1 warm-up + mean of {reps} steps, "
                     f"{dt:.2f} s/step; {torch.get_num_threads()} threads = the fastest setting on this host "
                     f"(profiles/r01_cpu_oracle_thread_sweep.txt)"}
    out.update(cpu_info())
    return out


class Scene:
    """Synthetic views resident in HBM: poses on the NeRF-synthetic sphere (or near-identity forward-facing poses)
    and random target images.  What the reference's loader would have put on the device."""

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
        self.images = [torch.rand(self.H, self.W, 3, generator=gen).to(dev) for _ in range(n_views)]
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
        ts = Dp.DepthTrainStep(args, kw, opt, grad_vars, distributed=world > 1)
        if force_dist and world == 1:
            ts.bucket = dp.GradientBucket(nets)
        torch.manual_seed(1000 + rank)     # this variant draws with torch.rand: decorrelate the ranks' draws

        def step(i):
            v = i % len(scene.poses)
            cols, target, pix = P.select_view_rays(scene.H, scene.W, scene.K, scene.poses[v], scene.images[v], a.rays,
                                                   scene.near, scene.far, seed=0, step=i, ray_id0=rank * a.rays,
                                                   want_pixels=True)
            target_h = scene.hyp[v][:, pix[:, 0].long(), pix[:, 1].long()].unsqueeze(-1)
            loss, _, _, _ = ts(cols.packed(), target, target_h)
            return loss
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
    return step, nets


def main():
    a = parse()
    ns, ni, desc = WORKLOADS[a.workload]
    a.n_samples = a.n_samples if a.n_samples is not None else ns
    a.n_importance = a.n_importance if a.n_importance is not None else ni
    import plnerf_amd as P
    from plnerf_amd import dp, functional as Fn
    rank, world, local = dp.init_from_env(force=a.force_dist)
    assert world == a.gpus, f"--gpus {a.gpus} but WORLD_SIZE={world}"
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist_on = world > 1 or a.force_dist

    scene = Scene(P, a.workload, a.views, dev)
    step, nets = build_step(P, a, a.precision, scene, dev, rank, world, a.force_dist)
    if a.force_dist and world == 1:     # exercise the RCCL path on one GPU
        _allreduce = dp.GradientBucket.allreduce_mean
        dp.GradientBucket.allreduce_mean = lambda self, group=None, force=False: _allreduce(self, group, True)
    R = a.rays
    rows_fine = R * (a.n_samples + a.n_importance)

    def sync():
        if dist_on:
            torch.distributed.barrier(device_ids=[local])
        torch.cuda.synchronize()

    def timed(step_fn, warmup, steps, timer=None):
        for i in range(warmup):
            step_fn(i)
        Fn.KERNEL_TIMER = timer
        sync()
        t0 = time.perf_counter()
        for i in range(warmup, warmup + steps):
            loss = step_fn(i)
        sync()
        dt = time.perf_counter() - t0
        Fn.KERNEL_TIMER = None
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        if dist_on:
            torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        return float(t.item()), float(loss.detach())

    timer = Fn.KernelTimer()
    dt, final_loss = timed(step, a.warmup, a.steps, timer)

    strict = None
    if a.precision != "fp32" and not a.no_strict_fp32 and world == 1:
        # the strictly reference-equal arithmetic (exact fp32 MFMA forward and backward), same workload, short leg
        step32, nets32 = build_step(P, a, "fp32", scene, dev, rank, world, False)
        s_steps = max(3, min(a.steps, 5))
        dt32, _ = timed(step32, 2, s_steps)
        strict = {"precision": "fp32", "ms_per_step": 1e3 * dt32 / s_steps, "rays_per_s": R * s_steps / dt32,
                  "steps": s_steps, "warmup": 2}
        del step32, nets32
        torch.cuda.empty_cache()

    if rank == 0:
        ms = 1e3 * dt / a.steps
        fwd_ms = timer.mean_ms(f"mlp_fwd[{rows_fine}]")
        bwd_ms = timer.mean_ms(f"mlp_bwd[{rows_fine}]")
        peak = PEAK_TFLOPS[a.precision]
        traffic = None
        fwd_kernel = FWD_KERNEL[a.precision]
        if a.workload == "depth_128_64" and a.precision in ("f16", "bf16"):
            # the caller-embedded 57|3 input of the depth variant: the plain modes have no register-resident variant for it
            fwd_kernel = "mlp_fwd_pp_kernel<1,true>"
        try:   # measured offline with rocprofv3 --pmc (cannot be collected from inside this process)
            t = json.load(open(os.path.join(ROOT, "profiles", "r02_traffic.json"))).get(a.precision)
            if t and t["rows_per_launch"] == rows_fine and t.get("kernel", fwd_kernel) == fwd_kernel:
                traffic = t["bytes"]
        except Exception:
            traffic = None
        ach = rows_fine * FWD_FLOP_PER_ROW / (fwd_ms * 1e-3) / 1e12 if fwd_ms else None
        h16 = a.precision != "fp32"
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
                                   f"choice + ray generation + render + backward + per-network grad all-reduce + Adam",
                       "global_rays": R * world, "precision": a.precision, "parallelism": f"dp{world}",
                       "rccl_world_size": torch.distributed.get_world_size() if torch.distributed.is_initialized() else 1,
                       "final_loss": final_loss},
            # the MLP is priced against the MFMA roofline on its ALGORITHMIC work,
            # 1,186,816 FLOP per network evaluation -- the 3 MFMA issues per product of the split modes are a cost,
            # not work.  The HBM view of the same launch (saved half planes written once) rides along.
            "roofline": {
                "bound": "mfma",
                "kernel": fwd_kernel + " (fine network, fused PE+12-layer MLP forward, saves backward state)",
                "achieved": ach, "peak": peak, "unit": "TFLOP/s", "frac": (ach / peak) if ach else None,
                "traffic": traffic, "launch_ms": fwd_ms, "rows_per_launch": rows_fine,
                "flop_per_row": FWD_FLOP_PER_ROW,
                "mfma_issued_tflops": (ach * MFMA_PER_PRODUCT[a.precision]) if ach else None,
                "mfma_issue_frac": (ach * MFMA_PER_PRODUCT[a.precision] / peak) if ach else None,
                "hbm_view": {"bytes_per_row": bytes_per_row, "achieved_gbs": ach_gbs, "peak_gbs": HBM_PEAK_GBS,
                             "frac": (ach_gbs / HBM_PEAK_GBS) if ach_gbs else None},
                "mlp_bwd_launch_ms": bwd_ms,
                "train_mlp_tflops": (rows_fine * TRAIN_FLOP_PER_ROW / ((fwd_ms + bwd_ms) * 1e-3) / 1e12)
                if (fwd_ms and bwd_ms) else None,
                "step_tflops": (R * (2 * a.n_samples + a.n_importance) * TRAIN_FLOP_PER_ROW / (ms * 1e-3) / 1e12),
            },
        }
        if strict is not None:
            out["strict_fp32"] = strict
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
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
