#!/bin/bash
# Training-forward launch time (the fine network's SAVE forward inside bench.py's step) per library variant.
R=${GRAFT_REPO_ROOT:-/root/repo}
export PLNERF_ALLOW_TOOLS_BUILD=1      # variant libraries carry ablation / trace switches
for r in 1 2; do
for lib in pp default $R/tools/_head/lib*.so; do
  unset PLNERF_HIP_LIB PLNERF_FWD_KERNEL
  if [ "$lib" = pp ]; then export PLNERF_FWD_KERNEL=pp; name=pp; elif [ "$lib" = default ]; then name=default; else export PLNERF_HIP_LIB=$lib; name=$(basename $lib .so); fi
  python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-strict-fp32 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']
print('$name', 'step', round(d['ms_per_step'], 3), 'ms  fwd(fine)', round(r['launch_ms'], 3), 'ms  bwd(fine)', round(r['mlp_bwd_launch_ms'], 3), 'ms  loss', round(d['config']['final_loss'], 6))"
done; done
