#!/bin/bash
export PLNERF_ALLOW_TOOLS_BUILD=1      # variant libraries carry ablation / trace switches (pl-nerf_amd/_lib.py refuses them otherwise)
# Same-box A/B of the working tree against a reference copy of the sources: GPU boxes differ by +-3 % in step
# time, so two variants are only comparable inside one gpurun call, alternating.
#   (in the build container)  rm -rf tools/_head && mkdir tools/_head && git archive HEAD pl-nerf_amd/csrc include | tar -x -C tools/_head
#   gpurun -- 'bash tools/ab_bench.sh'       then remove tools/_head again
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R/tools/_head/pl-nerf_amd/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -w -shared -o /tmp/a.so capi.hip quad.hip sampler.hip mlp_api.hip mlp_f32.hip mlp_bf16.hip
cd $R
for i in 1 2 3; do for v in a b; do
  if [ $v = a ]; then export PLNERF_HIP_LIB=/tmp/a.so; else unset PLNERF_HIP_LIB; fi
  python bench.py --no-cpu-baseline --steps 10 --warmup 3 ${BENCH_ARGS} 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$v (a = reference copy, b = working tree) step ms', round(d['ms_per_step'], 3), 'fwd', round(d['roofline']['launch_ms'],3), 'bwd ms', round(d['roofline']['mlp_bwd_launch_ms'], 3))"
done; done
