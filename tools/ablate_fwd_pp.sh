#!/bin/bash
export PLNERF_ALLOW_TOOLS_BUILD=1      # variant libraries carry ablation / trace switches (pl-nerf_amd/_lib.py refuses them otherwise)
# Timing experiments on the ping-pong forward (results wrong): PLNERF_ABLATE 16 = no LDS operand reads in the
# ring K loop, 32 = no weight refills from L2, 4 = no epilogue (conversion + LDS stores), 1 = n/a.
# Prints MLP-only inference throughput at 65536 x 192 per variant (the ping-pong kernel forced; needs the product's
# objects in csrc/: make first).
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R/pl-nerf_amd/csrc
for v in ${ABLATE_SET:-0 16 32 48}; do
  out=/tmp/libplnerf_ab${v}.so
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -w -DPLNERF_ABLATE=$v -c mlp_bf16.hip -o /tmp/mlp_bf16_ab$v.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $out capi.o quad.o sampler.o epilogue.o step.o mlp_api.o mlp_f32.o /tmp/mlp_bf16_ab$v.o mlp_rr.o mlp_rr_k_*.o
  PLNERF_FWD_KERNEL=pp PLNERF_HIP_LIB=$out python $R/tools/bench_mlp.py --precisions ${PRECS:-f16x3,bf16} --iters 5 2>/dev/null | grep inference | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('ABLATE=$v', d['precision'], round(d['ms'], 2), 'ms', round(d['tflops']), 'TFLOP/s')"
done
