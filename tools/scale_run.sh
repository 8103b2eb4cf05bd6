#!/bin/bash
# The 1 -> 8 GPU curve and its diagnosis in one JSON (see tools/scale_run.py):  bash tools/scale_run.sh [BENCH_rNN.json]
#   WORKLOAD=depth_128_64 bash tools/scale_run.sh      BASELINE configs[4] (the depth-supervised step) instead of configs[2]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
python $R/tools/scale_run.py ${1:+--bench-json $1} ${WORKLOAD:+--workload $WORKLOAD}
