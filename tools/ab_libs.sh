#!/bin/bash
# A/B of kernel variants built as separate libraries (tools/_head/lib*.so, git-ignored): the MLP-only benchmark per
# library, interleaved rounds in one gpurun call.  usage: bash tools/ab_libs.sh "<precisions>" <rounds> [bench_mlp args]
R=${GRAFT_REPO_ROOT:-/root/repo}
export PLNERF_ALLOW_TOOLS_BUILD=1      # variant libraries carry ablation / trace switches
prec=${1:-f16x3}; rounds=${2:-2}; shift 2
for r in $(seq $rounds); do
  for lib in default $R/tools/_head/lib*.so; do
    if [ "$lib" = default ]; then unset PLNERF_HIP_LIB; name=default; else export PLNERF_HIP_LIB=$lib; name=$(basename $lib .so); fi
    python $R/tools/bench_mlp.py --precisions $prec --iters 5 "$@" 2>/dev/null | grep '^{' | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('$name', d['precision'], d['what'][:32], round(d['ms'], 3), 'ms', round(d['tflops'], 1), 'TF')"
  done
done
