#!/bin/bash
# Whole-step A/B of library variants (tools/_head/lib*.so) against the product library: bench.py per variant,
# interleaved rounds, per precision.  usage: bash tools/ab_step.sh "<precisions>" <rounds>
R=${GRAFT_REPO_ROOT:-/root/repo}
export PLNERF_ALLOW_TOOLS_BUILD=1
for r in $(seq ${2:-2}); do for prec in ${1:-f16x3}; do
for lib in default $R/tools/_head/lib*.so; do
  unset PLNERF_HIP_LIB
  if [ "$lib" = default ]; then name=default; else export PLNERF_HIP_LIB=$lib; name=$(basename $lib .so); fi
  python $R/bench.py --precision $prec --steps 20 --warmup 5 --no-cpu-baseline --no-strict-fp32 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']
print('$name', '$prec', 'step', round(d['ms_per_step'], 3), 'ms  fwd(fine)', round(r['launch_ms'], 3), ' bwd(fine)', round(r['mlp_bwd_launch_ms'], 3), ' loss', round(d['config']['final_loss'], 6))"
done; done; done
