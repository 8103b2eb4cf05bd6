#!/bin/bash
# Same-box kernel-level A/B of variant libraries: rocprofv3 --kernel-trace --stats of one short bench.py run per library,
# top kernels by time.   gpurun -- 'bash tools/ab_libs_kstats.sh "bwdtm64 bwdtm192"'
R=${GRAFT_REPO_ROOT:-/root/repo}
export PLNERF_ALLOW_TOOLS_BUILD=1
cd /tmp && export TMPDIR=/tmp
for name in default $1; do
  if [ "$name" = default ]; then unset PLNERF_HIP_LIB; else export PLNERF_HIP_LIB=$R/tools/_head/lib$name.so; fi
  out=$R/gpurun_out/kstats_$name; rm -rf $out; mkdir -p $out
  rocprofv3 --kernel-trace --stats -d $out/p -o x -- python $R/bench.py --no-cpu-baseline --no-strict-fp32 --no-extra-legs --steps 20 --warmup 5 > $out/bench.json 2> $out/err.log
  db=$(find $out/p -name '*.db' | head -1)
  python $R/tools/rocpd_summary.py $db > $out/kernel_stats.csv 2>> $out/err.log
  rm -rf $out/p
  echo "== $name: $(python -c "import json;print(json.loads(open('$out/bench.json').read().strip().splitlines()[-1])['ms_per_step'])") ms/step under rocprof"
  head -9 $out/kernel_stats.csv | cut -c1-120
done
