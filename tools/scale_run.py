"""Multi-GPU first-contact kit: the whole 1 -> 8 GPU curve and its diagnosis from one command, for the
day a node exists.

    python tools/scale_run.py [--gpus 1,2,4,8] [--bench-json BENCH_r05.json] > scale.json

For every N with that many visible devices: `bench.py --gpus N --no-extra-legs` (the driver's command form: bench.py spawns
its own ranks; weak scaling, 4096 rays per GPU) and the bare latency of the step's collective (tools/allreduce_probe.py
under torch.distributed.run).  Writes ONE JSON object: per N rays/s, ms_per_step, the ranks' own step times (spread = a
slow rank or an exposed collective), the exposed all-reduce inside the step, the bare all-reduce, efficiency against N = 1;
and asserts that this run's N = 1 equals the recorded BENCH line within 3 % (a box that is off by more makes the curve
meaningless).  Budget for >= 7x at 8 GPUs: <= 12.5 % of the step exposed, i.e. <= 0.8 ms."""
import argparse, json, os, socket, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ap = argparse.ArgumentParser()
ap.add_argument("--gpus", default="1,2,4,8")
ap.add_argument("--steps", type=int, default=40)
ap.add_argument("--warmup", type=int, default=10)
ap.add_argument("--bench-json", default=None, help="the driver's BENCH_rNN.json (or a bench.py line) to check N = 1 against")
ap.add_argument("--workload", default="blender_64_128",
                help="bench.py's workload (depth_128_64 = BASELINE configs[4], the depth-supervised step, an 8-GPU configuration)")
a = ap.parse_args()
import torch
have = torch.cuda.device_count()


def free_port():
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def last_json(text):
    lines = [ln for ln in text.splitlines() if ln.startswith("{")]
    return json.loads(lines[-1]) if lines else None


env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
    env.pop(k, None)
out = {"what": f"training rays/s at N GPUs of one node, weak scaling (4096 rays per GPU), f16x3, workload {a.workload}; tools/scale_run.py",
       "visible_devices": have, "runs": {}}
for n in [int(x) for x in a.gpus.split(",")]:
    if n > have:
        out["runs"][str(n)] = {"skipped": f"needs {n} visible devices, this host has {have}"}
        continue
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", str(a.steps), "--warmup",
                        str(a.warmup), "--workload", a.workload, "--no-extra-legs", "--no-cpu-baseline", "--no-strict-fp32"] +
                       (["--force-dist"] if n == 1 else []),
                       env=env, text=True, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    line = last_json(r.stdout)
    if r.returncode != 0 or line is None:
        out["runs"][str(n)] = {"failed": r.returncode, "stderr_tail": r.stderr[-1500:]}
        continue
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
                        "--master-port", str(free_port()), os.path.join(ROOT, "tools", "allreduce_probe.py")],
                       env=env, text=True, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    out["runs"][str(n)] = {"rays_per_s": line["value"], "ms_per_step": line["ms_per_step"], "step_ms": line.get("step_ms"),
                           "ranks": line.get("ranks"), "rccl_world_size": line["config"].get("rccl_world_size"),
                           "backend": line["config"].get("backend"),
                           "allreduce_bare": last_json(p.stdout) or {"failed": p.returncode, "stderr_tail": p.stderr[-800:]}}
one = out["runs"].get("1", {})
if "rays_per_s" in one:
    for n, run in out["runs"].items():
        if "rays_per_s" in run:
            run["speedup_vs_1"] = run["rays_per_s"] / one["rays_per_s"]
            run["efficiency"] = run["speedup_vs_1"] / int(n)
            run["exposed_budget_ms_for_7x_at_8"] = 0.125 * one["ms_per_step"]
    if a.bench_json:
        rec = json.load(open(a.bench_json))
        rec = rec.get("parsed", rec) if isinstance(rec, dict) else rec      # (the driver's record keeps the line under "parsed")
        ref_ms = rec.get("ms_per_step")
        if ref_ms:
            out["n1_vs_recorded_bench"] = {"recorded_ms_per_step": ref_ms, "this_run_ms_per_step": one["ms_per_step"],
                                           "ratio": one["ms_per_step"] / ref_ms}
            assert abs(one["ms_per_step"] / ref_ms - 1.0) <= 0.03, \
                f"N = 1 is {one['ms_per_step']:.3f} ms/step here against the recorded {ref_ms:.3f}: this box is not the bench's box class"
print(json.dumps(out, indent=1))
