#!/bin/bash
# SQ / LDS / TCP counters per kernel of one bench.py run: three separate rocprofv3 --pmc passes (counters only; no trace
# domains), per kernel and grid (tools/pmc_summary.py) -> gpurun_out/pmc_sq/summary.txt
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/pmc_sq; rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --no-cpu-baseline --no-strict-fp32 --no-extra-legs --steps 3 --warmup 1 ${BENCH_ARGS}"
rocprofv3 --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU -d $out/p1 --output-format csv -- $B > $out/b1.json 2> $out/e1.log
rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d $out/p2 --output-format csv -- $B > $out/b2.json 2> $out/e2.log
rocprofv3 --pmc TCC_HIT TCC_MISS TCP_TCC_READ_REQ GRBM_GUI_ACTIVE -d $out/p3 --output-format csv -- $B > $out/b3.json 2> $out/e3.log
python $R/tools/pmc_summary.py $out/p1 $out/p2 $out/p3 > $out/summary.txt 2>> $out/e3.log
rm -rf $out/p1 $out/p2 $out/p3
grep -A14 -E "mlp_fwd_rr_kernel<2, true, false>.*grid=1572864|mlp_bwd_h16.*grid=3145728|wgrad_main" $out/summary.txt | head -60
