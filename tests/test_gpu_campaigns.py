"""The differential campaigns of tools/fuzz_*.py (hundreds of random configurations each, results under profiles/r05_fuzz_*)
as short legs of the GPU suite: a handful of cases per campaign with a seed of their own, exit code 0 = every stage inside its
bound on every ray.  Each tool states its bounds and what it compares with (the CPU oracle, or fp64 torch)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("tool,cases,extra", [
    ("fuzz_render_rays.py", 12, ()), ("fuzz_render_rays_depth.py", 8, ()), ("fuzz_train_step.py", 8, ()), ("fuzz_samplers.py", 40, ()),
    ("fuzz_quadrature.py", 40, ()), ("fuzz_mlp.py", 24, ()), ("fuzz_glue.py", 30, ()), ("fuzz_render_chunks.py", 6, ()),
    ("fuzz_train_step_depth.py", 8, ()),
    # (round 6) the depth-supervised step with half of its cases at is_joint=True; both training-step campaigns bound the gradient's
    # stages on the path's own inputs (tools/grad_stages.py) and report the end-to-end numbers
    ("fuzz_train_step_depth.py", 8, ("--joint", "0.5"))])
def test_campaign(tool, cases, extra):
    run = subprocess.run([sys.executable, os.path.join(ROOT, "tools", tool), "--cases", str(cases), "--seed", "2026", *extra],
                         capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert run.stdout.strip(), run.stderr[-2000:]
    out = json.loads(run.stdout.strip().splitlines()[-1])
    assert run.returncode == 0 and not out["violations"], (tool, out["violations"][:3], run.stderr[-500:])
