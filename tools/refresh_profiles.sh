#!/bin/bash
# One gpurun call that regenerates a round's evidence from the working tree:
#   gpurun --timeout 3000 -- 'ROUND=r06 COMMIT=<git rev-parse --short HEAD> bash tools/refresh_profiles.sh'
# then, in the build container:  for f in gpurun_out/r06p/*; do cp $f profiles/r06_$(basename $f); done
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/${ROUND:-r06}p
mkdir -p $out
cd $R
echo "${COMMIT:-unknown}" > $out/commit.txt
# the driver's command form: every leg on the one line (workloads, throughput mode, 65536x192 MLP, 800x800 frame, psnr_vs_ref, strict fp32, cpu)
python bench.py --steps 20 --warmup 5 > $out/bench_default_f16x3.json 2> $out/bench_default.err
for p in bf16x3 f16 bf16 fp32; do python bench.py --no-cpu-baseline --no-strict-fp32 --no-extra-legs --precision $p 2>/dev/null | tail -1; done > $out/bench_other_modes.jsonl
python bench.py --force-dist --no-cpu-baseline --no-strict-fp32 --no-extra-legs 2>/dev/null | tail -1 > $out/bench_force_dist_1gpu.json
python tools/scale_run.py --gpus 1,2 > $out/scale_run_on_1gpu_box.json 2> $out/scale_run.err
python tools/bench_stream_kernels.py 2>/dev/null | grep '^{' > $out/stream_kernels_262144rays.jsonl
cd /tmp && export TMPDIR=/tmp
# kernel-trace stats: the default workload and every other bench workload
for w in blender_64_128 blender_128_64 llff_ndc depth_128_64; do
  rocprofv3 --kernel-trace --stats -d $out/prof_$w -o x -- python $R/bench.py --workload $w --no-cpu-baseline --no-strict-fp32 --no-extra-legs > $out/${w}_bench_under_rocprof.json 2> $out/rocprof_$w.err
  db=$(find $out/prof_$w -name '*.db' | head -1)
  python $R/tools/rocpd_summary.py $db > $out/${w}_kernel_stats.csv 2>> $out/rocprof_$w.err
  rm -rf $out/prof_$w
done
cd $R && bash tools/pmc_traffic.sh > $out/pmc_traffic.log 2>&1; cp $R/gpurun_out/pmc_traffic/traffic.json $out/traffic.json; cp $R/gpurun_out/pmc_traffic/summary.txt $out/f16x3_hbm_traffic_pmc.txt
cd $R && bash tools/pmc_sq.sh > $out/pmc_sq.log 2>&1; cp $R/gpurun_out/pmc_sq/summary.txt $out/f16x3_pmc_sq_lds_tcp.txt
cd $R && bash tools/pmc_stream.sh > $out/pmc_stream.log 2>&1; cp $R/gpurun_out/pmc_stream/summary.txt $out/pmc_stream_kernels_262144rays.txt
python tools/soak.py --steps 3000 2>/dev/null | tail -1 > $out/soak.jsonl
tail -1 $out/bench_default_f16x3.json | cut -c1-300
head -12 $out/blender_64_128_kernel_stats.csv | cut -c1-150
