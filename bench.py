#!/usr/bin/env python
"""Headline benchmark: training rays/s of the PL-NeRF hot path on MI355X.

A "step" is one optimisation step of run_plnerf.py:1283-1316 on synthetic 800x800
Blender-style rays: render (coarse 64 + fine 64+128 samples, piecewise-linear quadrature,
exact PL importance sampling) -> mse(rgb)+mse(rgb0) -> backward through both MLPs ->
gradient all-reduce across ranks -> Adam on both networks.  N_rand = 4096 rays PER GPU
(BASELINE.json configs[1]; configs[2] is the same per-GPU load at 8 GPUs => weak scaling).

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

Rank 0 prints ONE JSON line.  `roofline` is measured live with HIP events around the
dominant kernel (the fused fine-network MLP forward) on the launch stream; `cpu_baseline`
times the CPU oracle (a port of the reference's PyTorch path) on the host cores on a bounded
sample.  The oracle is never on the measured GPU path.
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
HBM_PEAK_GBS = 8000.0                        # MI355X_MICROARCH.md: 8 TB/s HBM3E
# training forward, algorithmic bytes per row: saved state written + xyz read (12) + raw written (16)
FWD_TRAIN_BYTES_PER_ROW = {"fp32": 2596 * 4 + 28, "h16": 2528 * 2 + 272 + 28}   # fp32 planes | half planes + relu masks
PEAK_TFLOPS = {"fp32": 157.3, "bf16x3": 2500.0, "bf16": 2500.0, "f16x3": 2500.0, "f16": 2500.0}   # MI355X_MICROARCH.md, dense


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--rays", type=int, default=4096, help="N_rand per GPU")
    ap.add_argument("--n-samples", type=int, default=64)
    ap.add_argument("--n-importance", type=int, default=128)
    ap.add_argument("--precision", default="f16x3", choices=["fp32", "bf16x3", "bf16", "f16x3", "f16"],
                    help="MLP arithmetic: f16x3 (default) = 3-term IEEE-half split in the forward GEMMs + 3-term bf16 "
                         "split in the backward GEMMs on the 16-bit MFMA pipe, holds the 1e-5 parity bound with ~7x "
                         "margin; bf16x3 = bf16 split everywhere; fp32 = exact fp32 MFMA; bf16 / f16 = plain 16-bit "
                         "operands (throughput only)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-rays", type=int, default=1024)
    ap.add_argument("--cpu-threads", type=int, default=16)
    ap.add_argument("--force-dist", action="store_true",
                    help="initialise RCCL and run the gradient all-reduce even with one rank (path check)")
    return ap.parse_args()


def make_args(a, ckpt_dir):
    return Namespace(multires=10, i_embed=0, use_viewdirs=True, multires_views=4, N_importance=a.n_importance,
                     N_samples=a.n_samples, netdepth=8, netwidth=256, netdepth_fine=8, netwidth_fine=256,
                     netchunk=65536, lrate=5e-4, coarse_lrate=5e-4, ft_path=None, ckpt_dir=ckpt_dir, expname="exp",
                     no_reload=True, perturb=1.0, white_bkgd=True, raw_noise_std=0.0, mode="linear",
                     color_mode="midpoint", dataset="blender", no_ndc=False, lindisp=False,
                     precision=a.precision)


def cpu_baseline(a):
    """The oracle's training step on the host cores, bounded sample (1 warm-up + 2 steps)."""
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
    return {"value": n / dt, "unit": "rays/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{n} rays x ({a.n_samples}+{a.n_samples + a.n_importance}) samples, full train step "
                      f"(fwd+bwd+2xAdam), fp32 PyTorch CPU oracle, 1 warm-up + mean of {reps} steps, "
                      f"{dt:.2f} s/step"}


def main():
    a = parse()
    import plnerf_amd as P
    from plnerf_amd import dp, functional as Fn
    rank, world, local = dp.init_from_env(force=a.force_dist)
    assert world == a.gpus, f"--gpus {a.gpus} but WORLD_SIZE={world}"
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    ck = tempfile.mkdtemp()
    os.makedirs(os.path.join(ck, "exp"))
    torch.manual_seed(0)
    _stdout = sys.stdout
    sys.stdout = open(os.devnull, "w")          # create_nerf prints like the reference
    kw, _, _, _, opt, opt_c = P.create_nerf(make_args(a, ck), device=dev)
    sys.stdout = _stdout
    nets = [kw["network_fn"], kw["network_fine"]]
    dp.broadcast_parameters(nets)
    bucket = dp.GradientBucket(nets) if (world > 1 or a.force_dist) else None

    # every rank renders its own shard of the global batch: rays [rank*R, (rank+1)*R)
    R = a.rays
    batch_all, target_all, K = P.rays.synthetic_blender_rays(R * world, seed=0, device="cpu")
    lo, hi = dp.shard_rays(R * world, rank, world)
    rays = (batch_all[0, lo:hi].to(dev), batch_all[1, lo:hi].to(dev))
    target = target_all[lo:hi].to(dev)
    rows_fine = R * (a.n_samples + a.n_importance)
    rows_coarse = R * a.n_samples

    def step():
        rgb, disp, acc, extras = P.render(800, 800, K, chunk=32768, rays=rays, near=2.0, far=6.0, retraw=True,
                                          **kw)
        opt.zero_grad()
        opt_c.zero_grad()
        loss = P.img2mse(rgb, target) + P.img2mse(extras["rgb0"], target)
        loss.backward()
        if bucket is not None:
            bucket.allreduce_mean(force=a.force_dist)
        opt.step()
        opt_c.step()
        return loss

    def sync():
        if world > 1 or a.force_dist:
            torch.distributed.barrier(device_ids=[local])
        torch.cuda.synchronize()

    for _ in range(a.warmup):
        step()
    timer = Fn.KernelTimer()
    Fn.KERNEL_TIMER = timer
    sync()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        loss = step()
    sync()
    dt = time.perf_counter() - t0
    Fn.KERNEL_TIMER = None
    t = torch.tensor([dt], device=dev, dtype=torch.float64)
    if world > 1 or a.force_dist:
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
    dt = float(t.item())

    if rank == 0:
        ms = 1e3 * dt / a.steps
        fwd_ms = timer.mean_ms(f"mlp_fwd[{rows_fine}]")
        bwd_ms = timer.mean_ms(f"mlp_bwd[{rows_fine}]")
        peak = PEAK_TFLOPS[a.precision]
        traffic = None
        try:   # measured offline with rocprofv3 --pmc (cannot be collected from inside this process)
            t = json.load(open(os.path.join(ROOT, "profiles", "r01_traffic.json"))).get(a.precision)
            if t and t["rows_per_launch"] == rows_fine:
                traffic = t["bytes"]
        except Exception:
            traffic = None
        ach = rows_fine * FWD_FLOP_PER_ROW / (fwd_ms * 1e-3) / 1e12 if fwd_ms else None
        # Which roofline binds the training forward: 1,186,816 FLOP against the algorithmic bytes per row.
        # fp32 mode: 10,412 B (fp32 planes) = 114 FLOP/B against a ridge of 157.3 TF / 8 TB/s = 20 FLOP/B:
        # MFMA-bound.  16-bit modes: 5,356 B (half planes + relu masks) = 222 FLOP/B against a ridge of
        # 2.5 PF / 8 TB/s = 312 FLOP/B: HBM-bound.
        hbm_bound = a.precision != "fp32"
        bytes_per_row = FWD_TRAIN_BYTES_PER_ROW["h16" if hbm_bound else "fp32"]
        ach_gbs = rows_fine * bytes_per_row / (fwd_ms * 1e-3) / 1e9 if fwd_ms else None
        out = {
            "metric": "training rays/sec (coarse+fine, 64+128 samples)",
            "value": R * world * a.steps / dt, "unit": "rays/s", "n_gpus": world, "steps": a.steps,
            "warmup": a.warmup, "ms_per_step": ms, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": {"fp32": "f32", "bf16x3": "bf16x3 (3-term bf16 split, f32 accumulate)", "bf16": "bf16 (f32 accumulate)", "f16x3": "f16x3 (3-term f16 split, f32 accumulate)", "f16": "f16 (f32 accumulate)"}[a.precision], "data": "synthetic",
            "config": {"workload": f"BASELINE configs[1]: 800x800 Blender-style rays, N_rand={R}/GPU, "
                                   f"N_samples={a.n_samples}, N_importance={a.n_importance}, mode=linear/midpoint, "
                                   f"white_bkgd, perturb=1; full step = render + backward + grad all-reduce + 2xAdam",
                       "global_rays": R * world, "precision": a.precision, "parallelism": f"dp{world}",
                       "final_loss": float(loss.detach())},
            "roofline": {
                "bound": "hbm" if hbm_bound else "mfma",
                "kernel": {"fp32": "mlp_fwd_f32_kernel<2,true>", "f16x3": "mlp_fwd_pp_kernel<2,true>",
                           "f16": "mlp_fwd_pp_kernel<1,true>", "bf16x3": "mlp_fwd_pp_kernel<2,true>",
                           "bf16": "mlp_fwd_pp_kernel<1,true>"}[a.precision]
                          + " (fine network, fused PE+12-layer MLP forward, saves backward state)",
                "achieved": ach_gbs if hbm_bound else ach, "peak": HBM_PEAK_GBS if hbm_bound else peak,
                "unit": "GB/s" if hbm_bound else "TFLOP/s",
                "frac": ((ach_gbs / HBM_PEAK_GBS) if hbm_bound else (ach / peak)) if ach else None,
                "traffic": traffic, "launch_ms": fwd_ms, "rows_per_launch": rows_fine,
                "bytes_per_row": bytes_per_row, "flop_per_row": FWD_FLOP_PER_ROW,
                "mfma_tflops": ach, "mfma_peak_tflops": peak,
                # MFMA work actually issued: bf16x3 spends 3 MFMAs per algorithmic product
                "mfma_issue_frac": ((ach * {"fp32": 1, "bf16x3": 3, "bf16": 1, "f16x3": 3, "f16": 1}[a.precision] / peak) if ach else None),
                "mlp_bwd_launch_ms": bwd_ms,
                "train_mlp_tflops": (rows_fine * TRAIN_FLOP_PER_ROW / ((fwd_ms + bwd_ms) * 1e-3) / 1e12)
                if (fwd_ms and bwd_ms) else None,
            },
        }
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
