#!/bin/bash
# One gpurun call that regenerates round 3's evidence from the working tree:
#   gpurun --timeout 3000 -- 'COMMIT=<git rev-parse --short HEAD> bash tools/refresh_profiles_r03.sh'
# then, in the build container: copy gpurun_out/r03p/* over the matching profiles/r03_* files.
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/r03p
mkdir -p $out
cd $R
echo "${COMMIT:-unknown}" > $out/commit.txt
python bench.py --steps 20 --warmup 5 > $out/bench_default_f16x3.json 2> $out/bench_default.err
for w in blender_128_64 llff_ndc depth_128_64; do python bench.py --workload $w --steps 20 --warmup 5 --no-cpu-baseline --no-strict-fp32 2>/dev/null | tail -1; done > $out/bench_workloads.jsonl
for p in bf16x3 f16 bf16 fp32; do python bench.py --no-cpu-baseline --no-strict-fp32 --precision $p 2>/dev/null | tail -1; done > $out/bench_other_modes.jsonl
python bench.py --force-dist --no-cpu-baseline --no-strict-fp32 2>/dev/null | tail -1 > $out/bench_force_dist_1gpu.json
python bench.py --gpus 2 > $out/bench_gpus2_on_1gpu_box.txt 2>&1; echo "exit code $?" >> $out/bench_gpus2_on_1gpu_box.txt
python tools/bench_mlp.py --precisions fp32,f16x3,bf16x3,f16,bf16 --iters 5 2>/dev/null | grep '^{' > $out/mlp_only_65536x192.jsonl
python tools/bench_render.py 2>/dev/null | grep '^{' > $out/render_800x800_frame.jsonl
python tools/bench_render.py --precisions f16x3 --chunk 131072 2>/dev/null | grep '^{' >> $out/render_800x800_frame.jsonl
python tools/bench_stream_kernels.py 2>/dev/null | grep '^{' > $out/stream_kernels_262144rays.jsonl
cd /tmp && export TMPDIR=/tmp
# kernel-trace stats: the default workload and every other bench workload
for w in blender_64_128 blender_128_64 llff_ndc depth_128_64; do
  rocprofv3 --kernel-trace --stats -d $out/prof_$w -o x -- python $R/bench.py --workload $w --no-cpu-baseline --no-strict-fp32 > $out/${w}_bench_under_rocprof.json 2> $out/rocprof_$w.err
  db=$(find $out/prof_$w -name '*.db' | head -1)
  python $R/tools/rocpd_summary.py $db > $out/${w}_kernel_stats.csv 2>> $out/rocprof_$w.err
  rm -rf $out/prof_$w
done
rocprofv3 --kernel-trace --stats -d $out/prof2 -o x -- python $R/tools/bench_mlp.py --precisions f16x3 --iters 5 --train-rays 64 > /dev/null 2> $out/rocprof2.err
db=$(find $out/prof2 -name '*.db' | head -1)
python $R/tools/rocpd_summary.py $db 2>/dev/null | head -6 > $out/f16x3_mlp_inference_kernel_stats.csv
rm -rf $out/prof2
cd $R && bash tools/pmc_traffic.sh > $out/pmc_traffic.log 2>&1; cp $R/gpurun_out/pmc_traffic/traffic.json $out/traffic.json; cp $R/gpurun_out/pmc_traffic/summary.txt $out/f16x3_hbm_traffic_pmc.txt
cd $R && bash tools/pmc_sq.sh > $out/pmc_sq.log 2>&1; cp $R/gpurun_out/pmc_sq/summary.txt $out/f16x3_pmc_sq_lds_tcp.txt
tail -1 $out/bench_default_f16x3.json | cut -c1-300
head -8 $out/blender_64_128_kernel_stats.csv | cut -c1-150
cat $out/render_800x800_frame.jsonl | cut -c1-300
