#!/bin/bash
# One gpurun call: the GPU test suite, the default bench line, and the launcher's behaviour on a 1-GPU box.
#   gpurun --timeout 1500 -- 'bash tools/r03_check.sh'
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/r03; mkdir -p $out
cd $R
python -m pytest tests -m gpu -x -q -s > $out/tests.log 2>&1; echo "pytest rc=$?" | tee -a $out/tests.log
tail -5 $out/tests.log
python bench.py --steps 20 --warmup 5 > $out/bench_default_f16x3.json 2> $out/bench_default.err; echo "bench rc=$?"
tail -1 $out/bench_default_f16x3.json | cut -c1-1500
python bench.py --gpus 2 > $out/bench_gpus2.out 2> $out/bench_gpus2.err; echo "bench --gpus 2 rc=$? (2 expected on a 1-GPU box)"; cat $out/bench_gpus2.err | tail -2
python __graft_entry__.py smoke > $out/smoke.log 2>&1; echo "smoke rc=$?"; tail -3 $out/smoke.log
