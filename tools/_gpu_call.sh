set -x
mkdir -p gpurun_out
(timeout 1200 python -m pytest tests/test_gpu_modes.py tests/test_gpu_step.py -m gpu -q -s -x --timeout=1100 -k "input_gradients or two_ranks or eight_shards" > gpurun_out/r05_tests_misc.log 2>&1; echo "rc=$?" >> gpurun_out/r05_tests_misc.log)
grep -n "d/d\|passed\|failed\|Error\|max |param" gpurun_out/r05_tests_misc.log | cut -c1-300
