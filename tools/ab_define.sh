#!/bin/bash
export PLNERF_ALLOW_TOOLS_BUILD=1      # variant libraries carry ablation / trace switches (pl-nerf_amd/_lib.py refuses them otherwise)
# Same-box A/B of a compile-time definition with a value: DEF="PLNERF_WG_SPLITS=28" bash tools/ab_define.sh
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R/pl-nerf_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -w -D${DEF} -shared -o /tmp/def.so capi.hip quad.hip sampler.hip mlp_api.hip mlp_f32.hip mlp_bf16.hip
cd $R
for i in 1 2 3; do for v in off on; do
  if [ $v = on ]; then export PLNERF_HIP_LIB=/tmp/def.so; else unset PLNERF_HIP_LIB; fi
  python bench.py --no-cpu-baseline --steps 10 --warmup 3 ${BENCH_ARGS} 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('${DEF} $v: step ms', round(d['ms_per_step'], 3), 'fwd', round(d['roofline']['launch_ms'],3), 'bwd ms', round(d['roofline']['mlp_bwd_launch_ms'], 3), 'loss', d['config']['final_loss'])"
done; done
