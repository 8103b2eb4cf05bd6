#!/bin/bash
# Build experiment variants of the library on the GPU box and time the fused MLP forward (inference).
#   ABLATE_SET="0 2 4 ..."  PLNERF_ABLATE bit sets     WPF_SET="4 8"  weight prefetch depths
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R/pl-nerf_amd/csrc
for w in ${WPF_SET:-4}; do
for v in ${ABLATE_SET:-0}; do
  out=/tmp/libplnerf_ab${v}_w${w}.so
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -w -DPLNERF_ABLATE=$v -DPLNERF_WPF=$w -shared -o $out capi.hip quad.hip sampler.hip mlp_api.hip mlp_f32.hip mlp_bf16.hip
  echo "== ABLATE=$v WPF=$w"
  PLNERF_HIP_LIB=$out python $R/tools/bench_mlp.py --rays 32768 --precisions ${PRECS:-bf16x3,bf16} --train-rays ${TRAIN_RAYS:-256} 2>&1 | grep -E "${WHAT:-inference}"
done
done
