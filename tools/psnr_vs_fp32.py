"""'PSNR vs ref' at training scale: K steps at 4096 rays on the analytic scene in the benchmarked
arithmetic (f16x3: fp32-class forward, half-plane backward) and in the exact-fp32 kernels (gradients reference-equal to
1e-5), from identical weights, pixels and draws -- per seed of the draws, plus the comparison's NOISE FLOOR: the fp32 run
repeated from initial weights perturbed by 1e-7 relative.  One JSON line per seed, then a summary line.
    python tools/psnr_vs_fp32.py [--steps 2000] [--precision f16x3] [--rays 4096] [--seeds 0,1,2]"""
import argparse, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import plnerf_amd as P
from tools.scene import psnr_vs_ref

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=2000)
ap.add_argument("--rays", type=int, default=4096)
ap.add_argument("--precision", default="f16x3")
ap.add_argument("--views", type=int, default=8)
ap.add_argument("--seeds", default="0,1,2")
ap.add_argument("--no-twin", action="store_true")
ap.add_argument("--only-run", action="store_true", help="the precision's run alone (no fp32 reference, no twin)")
a = ap.parse_args()
rows = []
for seed in [int(x) for x in a.seeds.split(",")]:
    out = psnr_vs_ref(P, torch.device("cuda:0"), a.steps, rays=a.rays, precision=a.precision, views=a.views,
                      seed=seed, with_twin=not a.no_twin, only_run=a.only_run)
    out["seed"] = seed
    rows.append(out)
    print(json.dumps(out), flush=True)


if a.only_run:
    print(json.dumps({"summary": True, "only_run": True, "precision": a.precision, "lib": os.environ.get("PLNERF_HIP_LIB", "default"),
                      "psnr_train_tail_mean": [r["run"]["psnr_train_tail_mean"] for r in rows],
                      "psnr_heldout_view": [r["run"]["psnr_heldout_view"] for r in rows]}), flush=True)
    sys.exit(0)


def ms(v):
    return {"mean": float(np.mean(v)), "std": float(np.std(v)), "values": [float(x) for x in v]}


summary = {"summary": True, "precision": a.precision, "steps": a.steps, "rays": a.rays, "seeds": a.seeds,
           "gap_db_train": ms([r["gap_db_train"] for r in rows]),
           "gap_db_heldout": ms([r["gap_db_heldout"] for r in rows])}
if not a.no_twin:
    summary["noise_floor_db_train"] = ms([r["noise_floor_db_train"] for r in rows])
    summary["noise_floor_db_heldout"] = ms([r["noise_floor_db_heldout"] for r in rows])
print(json.dumps(summary), flush=True)
