"""Time the CPU oracle's training step at several torch thread counts (to pick an honest
cpu_baseline configuration for bench.py on the GPU box's host)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import plnerf_oracle as orc
n = 256
batch, target = orc.synthetic_blender_rays(n, seed=0)
kw = dict(N_samples=64, N_importance=128, mode="linear", color_mode="midpoint", perturb=1.0, white_bkgd=True,
          raw_noise_std=0.0)
print("cpu_count", os.cpu_count())
for th in (8, 16, 32, 64, 128):
    torch.set_num_threads(th)
    sd_c, sd_f = orc.closed_form_state_dict(0), orc.closed_form_state_dict(1)
    st = {}
    orc.train_step(sd_c, sd_f, batch, target, kw, adam_state=st)
    t0 = time.perf_counter()
    orc.train_step(sd_c, sd_f, batch, target, kw, adam_state=st)
    dt = time.perf_counter() - t0
    print(f"threads {th}: {dt:.2f} s/step, {n/dt:.1f} rays/s", flush=True)
