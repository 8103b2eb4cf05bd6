#!/bin/bash
# Per-kernel, per-grid stats of one bench.py run under rocprofv3 (first N lines of the CSV).
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/kstats; rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $out/prof -o x -- python $R/bench.py --no-cpu-baseline ${BENCH_ARGS} > $out/bench.json 2> $out/err.log
python $R/tools/rocpd_summary.py $(find $out/prof -name '*.db' | head -1) 2>/dev/null | head -${1:-14} | cut -c1-150
rm -rf $out/prof
