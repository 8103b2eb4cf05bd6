#!/bin/bash
# LDS counters per kernel of one bench.py run (own rocprofv3 pass, counters only).
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/pmc_lds; rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT -d $out/p1 --output-format csv -- python $R/bench.py --no-cpu-baseline --steps 3 --warmup 1 ${BENCH_ARGS} > $out/bench.json 2> $out/err.log
python $R/tools/pmc_summary.py $out/p1 | grep -A5 -E "bwd_h16|fwd_pp|wgrad_main|wgrad_thin" | head -60
rm -rf $out/p1
