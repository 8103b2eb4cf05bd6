#!/bin/bash
# (Re)compile csrc/mlp_rr.hip with extra flags and link a library variant.
#   bash tools/build_rr.sh                      -> resource report of all six instantiations (single translation unit)
#   bash tools/build_rr.sh <name> -DRR_...      -> tools/_head/librr_<name>.so (git-ignored A/B variant)
R=/root/repo; C=$R/pl-nerf_amd/csrc
name=$1; shift
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-function -mllvm -amdgpu-mfma-vgpr-form"
if [ -z "$name" ]; then
  cd $C && /opt/rocm/bin/hipcc $FLAGS -DRR_SINGLE_TU -c mlp_rr.hip -o /tmp/rr/v/mlp_rr_report.o -Rpass-analysis=kernel-resource-usage 2>&1 | grep -E "error|warning|ScratchSize|VGPRs Spill|AGPRs:|VGPRs:" | grep -v "Spill: 0\|Size \[bytes/lane\]: 0" | sort | uniq -c
else
  mkdir -p /tmp/rr/v $R/tools/_head
  cd $C && /opt/rocm/bin/hipcc $FLAGS "$@" -DRR_SINGLE_TU -c mlp_rr.hip -o /tmp/rr/v/mlp_rr_$name.o 2>&1 | grep -E "error" 
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/tools/_head/librr_$name.so $(ls *.o | grep -v "^mlp_rr\.o$\|^mlp_rr_k_") /tmp/rr/v/mlp_rr_$name.o
fi
