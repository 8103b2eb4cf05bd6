#!/bin/bash
# Same-box A/B: GPU boxes differ by 1-3 % in step time, so two variants are only comparable inside ONE gpurun call, with
# their legs interleaved.  One script for every kind of variant (it replaces round 1-5's ab_bench / ab_env / ab_flag /
# ab_fwd_train / ab_libs / ab_libs_kstats / ab_libs_step / ab_merged_bwd / ab_py / ab_step wrappers):
#
#   bash tools/ab.sh [-m step|mlp|kstats] [-r ROUNDS] [-o OUTFILE] "<variant> <variant> ..." [-- extra bench args]
#
# A variant is   default            the product library, no switch
#                lib:NAME           tools/_head/libNAME.so  (tools/build_variant.sh: one source recompiled with -D switches;
#                                   or a library built from another commit: git archive <rev> pl-nerf_amd/csrc include | make)
#                env:VAR=VALUE      an environment switch the Python package reads (PLNERF_MERGED_BWD=0, PLNERF_FWD_KERNEL=pp)
# Modes:         step    bench.py (30 steps after 8 warm-ups, no side legs): step, fine forward launch, both backwards, loss
#                mlp     tools/bench_mlp.py (the MLP alone; pass its arguments after --)
#                kstats  one short bench.py under rocprofv3 --kernel-trace --stats per variant, top kernels by time
#   gpurun --timeout 900 -- 'bash tools/ab.sh -r 4 -o gpurun_out/ab.txt "default lib:r05"'
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
export PLNERF_ALLOW_TOOLS_BUILD=1      # variant libraries may carry trace switches or an older ABI number (pl-nerf_amd/_lib.py)
mode=step; rounds=3; out=/dev/null
while getopts "m:r:o:" o; do case $o in m) mode=$OPTARG;; r) rounds=$OPTARG;; o) out=$OPTARG;; esac; done
shift $((OPTIND - 1))
variants=$1; shift; [ "$1" = "--" ] && shift
[ "$out" != /dev/null ] && echo "# tools/ab.sh -m $mode -r $rounds \"$variants\" $*   (same box, interleaved)" > $out
run() {      # $1 = variant: sets the environment, runs the mode's command, prints one line
  local v=$1 name=$1
  unset PLNERF_HIP_LIB
  case $v in
    default) ;;
    lib:*) export PLNERF_HIP_LIB=$R/tools/_head/lib${v#lib:}.so ;;
    env:*) export "${v#env:}" ;;
  esac
  shift
  case $mode in
    step)
      python $R/bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-strict-fp32 --no-extra-legs "$@" 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); rf = d['roofline']
print('%-28s step %.3f ms (min %.3f median %.3f)  fine fwd %.3f ms  bwd %.3f ms  loss %.7f' % ('$name', d['ms_per_step'], d['step_ms']['min'], d['step_ms']['median'], rf['launch_ms'], (rf.get('mlp_bwd_both_networks_ms') or rf.get('mlp_bwd_launch_ms') or 0.0), d['config']['final_loss']))" ;;
    mlp)
      python $R/tools/bench_mlp.py --iters 5 "$@" 2>/dev/null | grep '^{' | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('%-28s' % '$name', d['precision'], d['what'][:32], round(d['ms'], 3), 'ms', round(d['tflops'], 1), 'TF')" ;;
    kstats)
      local o=$R/gpurun_out/kstats_$(echo $name | tr ':=/' '___'); rm -rf $o; mkdir -p $o
      (cd /tmp && TMPDIR=/tmp rocprofv3 --kernel-trace --stats -d $o/p -o x -- python $R/bench.py --no-cpu-baseline --no-strict-fp32 --no-extra-legs --steps 20 --warmup 5 "$@" > $o/bench.json 2> $o/err.log)
      python $R/tools/rocpd_summary.py $(find $o/p -name '*.db' | head -1) > $o/kernel_stats.csv 2>> $o/err.log; rm -rf $o/p
      echo "== $name: $(python -c "import json;print(json.loads(open('$o/bench.json').read().strip().splitlines()[-1])['ms_per_step'])") ms/step under rocprof"
      head -12 $o/kernel_stats.csv | cut -c1-130 ;;
  esac
  case $v in env:*) local kv=${v#env:}; unset "${kv%%=*}" ;; esac
}
for r in $(seq $rounds); do
  for v in $variants; do run $v "$@" | sed "s/^/round $r  /" | tee -a $out; done
  [ $mode = kstats ] && break      # (one profile per variant)
done
