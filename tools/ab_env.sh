#!/bin/bash
# Same-box A/B of an environment switch read by the library: alternates bench.py with and without $VAR=1.
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
for i in 1 2 3; do for v in off on; do
  if [ $v = on ]; then export $VAR=1; else unset $VAR; fi
  python bench.py --no-cpu-baseline --steps 10 --warmup 3 ${BENCH_ARGS} 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$VAR $v: step ms', round(d['ms_per_step'], 3), 'fwd', round(d['roofline']['launch_ms'],3), 'bwd ms', round(d['roofline']['mlp_bwd_launch_ms'], 3), 'loss', d['config']['final_loss'])"
done; done
