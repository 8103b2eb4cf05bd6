#!/bin/bash
# SQ / LDS / TCP counters of the per-ray (HBM / VALU bound) kernels at a chip-filling size (262,144 rays): separate
# rocprofv3 --pmc passes, counters only (no trace domains) -> gpurun_out/pmc_stream/summary.txt
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/pmc_stream; rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
B="python $R/tools/bench_stream_kernels.py --rays 262144 --reps 2"
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES -d $out/p1 --output-format csv -- $B > $out/b1.json 2> $out/e1.log
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU -d $out/p2 --output-format csv -- $B > $out/b2.json 2> $out/e2.log
rocprofv3 --pmc TCP_TCC_READ_REQ TCC_HIT TCC_MISS GRBM_GUI_ACTIVE SQ_WAVES -d $out/p3 --output-format csv -- $B > $out/b3.json 2> $out/e3.log
rocprofv3 --pmc SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INSTS_TEX_LOAD SQ_INSTS_TEX_STORE SQ_INSTS_VALU_FLOPS_FP64 -d $out/p4 --output-format csv -- $B > $out/b4.json 2> $out/e4.log
PMC_FILTER="epilogue|merge_sort|sample_pl|quad_|coarse_samples|ray_points|stratified" python $R/tools/pmc_summary.py $out/p1 $out/p2 $out/p3 $out/p4 > $out/summary.txt 2>> $out/e4.log
rm -rf $out/p1 $out/p2 $out/p3 $out/p4
head -150 $out/summary.txt
