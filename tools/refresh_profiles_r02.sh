#!/bin/bash
# One gpurun call that regenerates round 2's evidence from the working tree:
#   gpurun --timeout 2400 -- 'bash tools/refresh_profiles_r02.sh'
# then, in the build container: copy gpurun_out/r02/* over the matching profiles/r02_* files.
# (The tile-walk trace needs tools/_head/librr_trace.so: `bash tools/build_rr.sh trace -DRR_TRACE=100` before the call;
#  tools/pmc_sq.sh, tools/soak.py, tools/train_curve.py and tools/ablate_rr.sh produce the remaining r02 files.)
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/r02
mkdir -p $out
cd $R
python bench.py --steps 20 --warmup 5 > $out/bench_default_f16x3.json 2> $out/bench_default.err
for w in blender_128_64 llff_ndc depth_128_64; do python bench.py --workload $w --no-cpu-baseline --no-strict-fp32 2>/dev/null | tail -1; done > $out/bench_workloads.jsonl
for p in bf16x3 f16 bf16 fp32; do python bench.py --no-cpu-baseline --no-strict-fp32 --precision $p 2>/dev/null | tail -1; done > $out/bench_other_modes.jsonl
python bench.py --force-dist --no-cpu-baseline --no-strict-fp32 2>/dev/null | tail -1 > $out/bench_force_dist_1gpu.json
python tools/bench_mlp.py --precisions fp32,f16x3,bf16x3,f16,bf16 --iters 5 2>/dev/null | grep '^{' > $out/mlp_only_65536x192.jsonl
PLNERF_FWD_KERNEL=pp python tools/bench_mlp.py --precisions f16x3,f16 --iters 5 2>/dev/null | grep '^{' > $out/mlp_only_65536x192_pingpong_kernel.jsonl
python tools/bench_render.py 2>/dev/null | grep '^{' > $out/render_800x800_frame.jsonl
python tools/bench_stream_kernels.py 2>/dev/null | grep '^{' > $out/stream_kernels_262144rays.jsonl
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $out/prof -o x -- python $R/bench.py --no-cpu-baseline --no-strict-fp32 > $out/f16x3_bench_under_rocprof.json 2> $out/rocprof.err
db=$(find $out/prof -name '*.db' | head -1)
python $R/tools/rocpd_summary.py $db > $out/f16x3_bench_kernel_stats.csv 2> $out/summary.err
rm -rf $out/prof
rocprofv3 --kernel-trace --stats -d $out/prof2 -o x -- python $R/tools/bench_mlp.py --precisions f16x3 --iters 5 --train-rays 64 > /dev/null 2> $out/rocprof2.err
db=$(find $out/prof2 -name '*.db' | head -1)
python $R/tools/rocpd_summary.py $db 2>/dev/null | head -6 > $out/f16x3_mlp_inference_kernel_stats.csv
rm -rf $out/prof2
PLNERF_ALLOW_TOOLS_BUILD=1 PLNERF_HIP_LIB=$R/tools/_head/librr_trace.so python $R/tools/trace_rr.py f16x3 f16 2>/dev/null | grep cycles > $out/rr_tile_walk_clock.txt
cd $R && bash tools/pmc_traffic.sh > $out/pmc_traffic.log 2>&1; cp $R/gpurun_out/pmc_traffic/traffic.json $out/traffic.json; cp $R/gpurun_out/pmc_traffic/summary.txt $out/f16x3_hbm_traffic_pmc.txt
tail -1 $out/bench_default_f16x3.json | cut -c1-300
head -8 $out/f16x3_bench_kernel_stats.csv | cut -c1-150
cat $out/stream_kernels_262144rays.jsonl | cut -c1-200
