#!/bin/bash
# The differential campaigns of tools/fuzz_*.py at their full case counts (round 5's counts and seeds), one JSON per
# campaign under gpurun_out/<ROUND>_fuzz/:   gpurun --timeout 5400 -- 'ROUND=r06 bash tools/run_campaigns.sh [tool ...]'
# then in the build container:  for f in gpurun_out/r06_fuzz/*.json; do cp $f profiles/r06_$(basename $f); done
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
out=$R/gpurun_out/${ROUND:-r06}_fuzz; mkdir -p $out
declare -A CASES=([render_rays]=300 [render_rays_depth]=150 [train_step]=120 [train_step_depth]=120 [mlp]=200 [render_chunks]=60 [samplers]=400 [quadrature]=400 [glue]=300)
declare -A SEED=([render_rays]=7 [render_rays_depth]=11 [train_step]=5 [train_step_depth]=31 [mlp]=21 [render_chunks]=17 [samplers]=3 [quadrature]=9 [glue]=13)
if [ "${SEEDSET:-1}" = 2 ]; then      # round 5's second pass: other seeds, more cases (profiles/r05_fuzz_second_seeds.json)
  CASES=([render_rays]=600 [render_rays_depth]=300 [train_step]=200 [train_step_depth]=200 [mlp]=400 [render_chunks]=90 [samplers]=600 [quadrature]=600 [glue]=400)
  SEED=([render_rays]=8 [render_rays_depth]=12 [train_step]=6 [train_step_depth]=32 [mlp]=22 [render_chunks]=18 [samplers]=4 [quadrature]=10 [glue]=14)
  out=${out}_seeds2; mkdir -p $out
fi
if [ "${SEEDSET:-1}" -ge 3 ]; then      # round 6's further passes: the second pass's counts, its seeds + 100 (SEEDSET - 2): 108, 112, ... / 208, 212, ...
  CASES=([render_rays]=600 [render_rays_depth]=300 [train_step]=200 [train_step_depth]=200 [mlp]=400 [render_chunks]=90 [samplers]=600 [quadrature]=600 [glue]=400)
  o=$(( 100 * (SEEDSET - 2) ))
  SEED=([render_rays]=$((8 + o)) [render_rays_depth]=$((12 + o)) [train_step]=$((6 + o)) [train_step_depth]=$((32 + o)) [mlp]=$((22 + o)) [render_chunks]=$((18 + o)) [samplers]=$((4 + o)) [quadrature]=$((10 + o)) [glue]=$((14 + o)))
  out=${out}_seeds${SEEDSET}; mkdir -p $out
fi
declare -A EXTRA=()
[ "${SEEDSET:-1}" -ge 5 ] && EXTRA[train_step_depth]="--joint 0.5"      # (from the fifth pass on: half of the depth-supervised steps with is_joint=True)
tools=${@:-render_rays render_rays_depth train_step train_step_depth mlp render_chunks}
cd $R
for t in $tools; do
  s=$(date +%s)
  timeout ${CAMPAIGN_TIMEOUT:-1500} python tools/fuzz_$t.py --cases ${CASES[$t]} --seed ${SEED[$t]} ${EXTRA[$t]} 2> $out/fuzz_$t.err | tail -1 > $out/fuzz_$t.json
  echo "fuzz_$t: rc=${PIPESTATUS[0]} $(( $(date +%s) - s )) s  $(python -c "import json,sys; d=json.load(open('$out/fuzz_$t.json')); print('violations', len(d['violations']))" 2>&1 | tail -1)"
done
