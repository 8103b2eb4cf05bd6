#!/bin/bash
# Build a kernel-variant library for a same-box A/B: ONE source file recompiled with extra -D switches, linked with the tree's
# other objects (run `make -C pl-nerf_amd/csrc` first), written to tools/_head/lib<name>.so (git-ignored; it travels with gpurun).
#   bash tools/build_variant.sh rrPFD3 mlp_rr_k_2_train.hip "-DRR_PFD=3"
#   gpurun -- 'bash tools/ab.sh "default lib:rrPFD3"'
# (how profiles/r05_forward_knobs_ab.txt and r05_dgrad_prefetch_ab.txt were made; replaces tools/ab_define.sh, whose
# hard-wired source list predated the split of csrc/ into per-kernel files)
set -e
name=$1; src=$2; defs=$3
R=$(cd "$(dirname "$0")/.." && pwd)
cd $R/pl-nerf_amd/csrc
extra=""
case $src in mlp_rr*) extra="-mllvm -amdgpu-mfma-vgpr-form";; esac      # (as the Makefile does for the register-resident forward)
mkdir -p $R/tools/_head
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -w $extra $defs -c $src -o /tmp/variant_$name.o \
    -Rpass-analysis=kernel-resource-usage 2> /tmp/variant_$name.err || { tail -5 /tmp/variant_$name.err; exit 1; }
objs=$(ls *.o | grep -v "^${src%.hip}.o$" | tr '\n' ' ')
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/tools/_head/lib$name.so $objs /tmp/variant_$name.o
grep -E "Function Name|VGPRs:|VGPRs Spill|ScratchSize" /tmp/variant_$name.err | paste - - - - | sed 's/\[-Rpass-analysis=kernel-resource-usage\]//g' | cut -c1-220 | head -12
echo "built tools/_head/lib$name.so"
