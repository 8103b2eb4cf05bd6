#!/bin/bash
# What each ingredient of the register-resident forward (csrc/mlp_rr_body.inc) costs: the kernel rebuilt with RR_ABLATE bits
# (results WRONG by construction; tools builds only) and timed against the product library in one gpurun call.
#   here:        bash tools/ablate_rr.sh build     -> tools/_head/librr_abl_{nodma,nolds,nobar,noepi,all}.so
#   on the GPU:  bash tools/ablate_rr.sh           -> gpurun_out/rr_ablation.txt
R=${GRAFT_REPO_ROOT:-/root/repo}
if [ "$1" = build ]; then
  for v in nodma:1 nolds:2 nobar:4 noepi:8 all:15; do bash $R/tools/build_rr.sh abl_${v%%:*} -DRR_ABLATE=${v##*:} & done; wait
  exit 0
fi
out=$R/gpurun_out/rr_ablation.txt; mkdir -p $R/gpurun_out
{
  echo "# mlp_fwd_rr_kernel, inference, 65,536 x 192 rows; RR_ABLATE variants: nodma = no weight DMA after the prologue,"
  echo "# nolds = no weight-fragment LDS reads, nobar = no barrier at the hand-over points, noepi = no epilogue arithmetic,"
  echo "# all = the four together (what is left: the MFMA stream, the encoding prologue and the output stores)"
  bash $R/tools/ab_libs.sh f16x3,f16 2 --train-rays 0 | grep -v "abl.*training\|trace"
} > $out 2>&1
cat $out
