set -x
COMMIT=a44181b bash tools/refresh_profiles_r05.sh > gpurun_out/r05_refresh.log 2>&1
tail -5 gpurun_out/r05_refresh.log
(timeout 2400 python -m pytest tests -m gpu -q -x --timeout=2000 > gpurun_out/r05_tests_final.log 2>&1; echo "rc=$?" >> gpurun_out/r05_tests_final.log)
tail -4 gpurun_out/r05_tests_final.log
