#!/bin/bash
# Same-box A/B of train.TrainStep's schedules (bench.py --pipeline 0 | 1 | 2), interleaved rounds:
#   gpurun --timeout 900 -- 'bash tools/ab_pipeline.sh [workload] [rounds]'
R=${GRAFT_REPO_ROOT:-/root/repo}
w=${1:-blender_64_128}
rounds=${2:-3}
out=$R/gpurun_out/ab_pipeline
mkdir -p $out
cd $R
: > $out/${w}.txt
for r in $(seq 1 $rounds); do
  for p in 0 1 2; do
    python bench.py --workload $w --pipeline $p --steps 40 --warmup 10 --no-cpu-baseline --no-strict-fp32 2>/dev/null | tail -1 | \
      python -c "import sys,json; d=json.loads(sys.stdin.read()); rf=d['roofline']; print('round $r pipeline $p: %.3f ms/step  step_ms min %.3f median %.3f max %.3f  fine fwd launch %.3f ms  fine bwd %.3f ms  loss %.6f' % (d['ms_per_step'], d['step_ms']['min'], d['step_ms']['median'], d['step_ms']['max'], rf['launch_ms'], rf['mlp_bwd_launch_ms'], d['config']['final_loss']))" >> $out/${w}.txt
  done
done
cat $out/${w}.txt
