#!/bin/bash
# The 1 -> 8 GPU curve and its diagnosis in one JSON (see tools/scale_run.py):  bash tools/scale_run.sh [BENCH_rNN.json]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
python $R/tools/scale_run.py ${1:+--bench-json $1}
