#!/bin/bash
# Timing experiments on the training forward's copy waves (results wrong): PLNERF_ABLATE 256 = no plane
# stores, 512 = no relu-mask stores, 1024 = copy waves idle.  Prints launch times from bench.py.
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R/pl-nerf_amd/csrc
for v in ${ABLATE_SET:-0 256 512 768 1024}; do
  out=/tmp/libplnerf_ab${v}.so
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -w -DPLNERF_ABLATE=$v -shared -o $out capi.hip quad.hip sampler.hip mlp_api.hip mlp_f32.hip mlp_bf16.hip
  for p in ${PRECS:-f16x3}; do
    PLNERF_HIP_LIB=$out python $R/bench.py --no-cpu-baseline --steps 5 --warmup 2 --precision $p 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('ABLATE=$v', d['config']['precision'], 'step ms', round(d['ms_per_step'], 3), 'fine fwd ms', round(d['roofline']['launch_ms'], 3), 'bwd ms', round(d['roofline']['mlp_bwd_launch_ms'], 3))"
  done
done
