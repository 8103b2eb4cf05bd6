#!/bin/bash
# One gpurun call that regenerates the round's headline evidence from the working tree:
#   gpurun --timeout 2400 -- 'bash tools/refresh_profiles.sh r18'
# then, in the build container: copy gpurun_out/<tag>/* over the matching profiles/r01_* files.
tag=${1:-rNN}
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/$tag
mkdir -p $out
cd $R
python bench.py > $out/bench_default.json 2> $out/bench_default.err
for p in bf16x3 f16 bf16 fp32; do python bench.py --no-cpu-baseline --precision $p 2>/dev/null | tail -1; done > $out/bench_other_modes.jsonl
python tools/bench_mlp.py --precisions fp32,f16x3,bf16x3,f16,bf16 --iters 5 2>/dev/null | grep '^{' > $out/mlp_only.jsonl
python tools/bench_depth.py 2>/dev/null | grep '^{' > $out/depth.jsonl
python tools/bench_render.py 2>/dev/null | grep '^{' > $out/render.jsonl
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $out/prof -o x -- python $R/bench.py --no-cpu-baseline > $out/bench_under_rocprof.json 2> $out/rocprof.err
db=$(find $out/prof -name '*.db' | head -1)
python $R/tools/rocpd_summary.py $db > $out/kernel_stats.csv 2> $out/summary.err
rm -rf $out/prof
tail -1 $out/bench_default.json | cut -c1-400
head -12 $out/kernel_stats.csv | cut -c1-160
