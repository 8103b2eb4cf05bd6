#!/bin/bash
# Same-box A/B of kernel-variant libraries (tools/_head/lib<name>.so, git-ignored; built with a -D switch) on the full
# training step: bench.py per library, interleaved rounds in one gpurun call.
#   gpurun --timeout 900 -- 'bash tools/ab_libs_step.sh "bwdtm128" 3 [bench args]'
R=${GRAFT_REPO_ROOT:-/root/repo}
export PLNERF_ALLOW_TOOLS_BUILD=1
names=${1:-}; rounds=${2:-3}; shift 2
for r in $(seq $rounds); do
  for name in default $names; do
    if [ "$name" = default ]; then unset PLNERF_HIP_LIB; else export PLNERF_HIP_LIB=$R/tools/_head/lib$name.so; fi
    python $R/bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-strict-fp32 --no-extra-legs "$@" 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); rf = d['roofline']
print('round $r %-10s step %.3f ms  (min %.3f median %.3f)  fine fwd %.3f ms  bwd %.3f ms  loss %.6f' % ('$name', d['ms_per_step'], d['step_ms']['min'], d['step_ms']['median'], rf['launch_ms'], (rf.get('mlp_bwd_both_networks_ms') or rf.get('mlp_bwd_launch_ms') or 0.0), d['config']['final_loss']))"
  done
done
