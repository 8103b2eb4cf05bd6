set -x
mkdir -p gpurun_out
(timeout 1500 python -m pytest tests/test_gpu_step.py -m gpu -q -s -x --timeout=1400 -k "merged or reproducible or workspace_held or train_step_from_a_view or two_ranks" > gpurun_out/r05_tests_merged.log 2>&1; echo "rc=$?" >> gpurun_out/r05_tests_merged.log)
tail -12 gpurun_out/r05_tests_merged.log
timeout 900 bash tools/ab_merged_bwd.sh blender_64_128 4 > /dev/null 2>&1
cat gpurun_out/r05_merged_bwd_ab_blender_64_128.txt
(timeout 2000 python -m pytest tests/test_gpu_fullsize.py -m gpu -q -s --timeout=1900 > gpurun_out/r05_tests_fullsize.log 2>&1; echo "rc=$?" >> gpurun_out/r05_tests_fullsize.log)
tail -5 gpurun_out/r05_tests_fullsize.log
