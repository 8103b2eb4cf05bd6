#!/bin/bash
# Build a trace variant of the library (mlp_bf16.hip recompiled with -DPLNERF_TRACE=<block>, linked with the product's
# other objects -- run `make -C pl-nerf_amd/csrc` first) and print the phase breakdown of the ping-pong forward kernel
# (BWD=1: of the dgrad kernel).  Works on the GPU box (hipcc is in the image) or here (then ship tools/_head/).
export PLNERF_ALLOW_TOOLS_BUILD=1      # variant libraries carry ablation / trace switches (pl-nerf_amd/_lib.py refuses them otherwise)
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R/pl-nerf_amd/csrc
out=/tmp/libplnerf_trace.so
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -w -DPLNERF_TRACE=${TRACE_BLOCK:-3000} ${EXTRA_DEFS} -c mlp_bf16.hip -o /tmp/mlp_bf16_trace.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $out capi.o quad.o sampler.o epilogue.o step.o mlp_api.o mlp_f32.o /tmp/mlp_bf16_trace.o mlp_rr.o mlp_rr_k_*.o
if [ -n "$BWD" ]; then PLNERF_HIP_LIB=$out python $R/tools/trace_bwd.py; exit 0; fi
for p in ${PRECS:-bf16 bf16x3}; do
  for m in ${MODES:-inference train}; do PLNERF_FWD_KERNEL=pp PLNERF_HIP_LIB=$out python $R/tools/trace_fwd.py $p $m; done
done
