#!/bin/bash
# Same-box A/B of train.TrainStep's merged backward (both networks' backward in one launch sequence + max |g_raw| as
# plnerf_quad_bwd's by-product) against the autograd order (PLNERF_MERGED_BWD=0): interleaved rounds of bench.py.
#   gpurun --timeout 900 -- 'bash tools/ab_merged_bwd.sh [workload] [rounds]'
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
w=${1:-blender_64_128}; rounds=${2:-4}
out=$R/gpurun_out/r05_merged_bwd_ab_$w.txt
echo "# bench.py --workload $w --steps 40 --warmup 10, f16x3, same box, interleaved; PLNERF_MERGED_BWD = 0 (autograd order: two backwards, two absmax passes) | 1 (one launch sequence)" > $out
for r in $(seq 1 $rounds); do for m in 0 1; do
  PLNERF_MERGED_BWD=$m python bench.py --workload $w --steps 40 --warmup 10 --no-cpu-baseline --no-strict-fp32 --no-extra-legs 2>/dev/null | tail -1 | \
    python -c "import sys,json; d=json.loads(sys.stdin.read()); rf=d['roofline']; print('round $r merged $m: %.3f ms/step  step_ms min %.3f median %.3f max %.3f  fine fwd %.3f ms  bwd fine %s both %s  loss %.7f' % (d['ms_per_step'], d['step_ms']['min'], d['step_ms']['median'], d['step_ms']['max'], rf['launch_ms'], rf['mlp_bwd_launch_ms'], rf['mlp_bwd_both_networks_ms'], d['config']['final_loss']))" >> $out
done; done
cat $out
