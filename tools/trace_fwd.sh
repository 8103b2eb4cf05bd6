#!/bin/bash
export PLNERF_ALLOW_TOOLS_BUILD=1      # variant libraries carry ablation / trace switches (pl-nerf_amd/_lib.py refuses them otherwise)
# Build a trace variant of the library on the GPU box and print the forward kernel's phase breakdown.
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R/pl-nerf_amd/csrc
out=/tmp/libplnerf_trace.so
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -w -DPLNERF_TRACE=${TRACE_BLOCK:-3000} -shared -o $out capi.hip quad.hip sampler.hip mlp_api.hip mlp_f32.hip mlp_bf16.hip
if [ -n "$BWD" ]; then PLNERF_HIP_LIB=$out python $R/tools/trace_bwd.py; exit 0; fi
for p in ${PRECS:-bf16 bf16x3}; do
  for m in ${MODES:-inference train}; do PLNERF_HIP_LIB=$out python $R/tools/trace_fwd.py $p $m; done
done
