#!/bin/bash
# Same-box A/B of the working tree against a copy of HEAD's package (python + kernels) under tools/_head:
#   rm -rf tools/_head && mkdir tools/_head && git archive HEAD pl-nerf_amd include | tar -x -C tools/_head
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
make -s -C $R/tools/_head/pl-nerf_amd/csrc > /dev/null 2>&1
mkdir -p /tmp/headtree && rm -rf /tmp/headtree/* && cp -r $R/tools/_head/pl-nerf_amd /tmp/headtree/ && cp -r $R/oracle $R/profiles $R/bench.py $R/plnerf_amd.py /tmp/headtree/
cd $R
for i in 1 2 3; do for v in a b; do
  if [ $v = a ]; then d=/tmp/headtree; else d=$R; fi
  (cd $d && python bench.py --no-cpu-baseline --steps 10 --warmup 3 ${BENCH_ARGS} 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$v (a = HEAD, b = working tree) step ms', round(d['ms_per_step'], 3), 'loss', d['config']['final_loss'])")
done; done
