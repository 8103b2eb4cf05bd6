set -x
mkdir -p gpurun_out
(timeout 2400 python -m pytest tests -m gpu -q -x --timeout=2000 > gpurun_out/r05_tests_all2.log 2>&1; echo "rc=$?" >> gpurun_out/r05_tests_all2.log)
tail -6 gpurun_out/r05_tests_all2.log
bash tools/ab_libs_step.sh "plnerf_hip_5ccccda" 4 > gpurun_out/r05_enc_planes_tiled_ab.txt 2>&1
cat gpurun_out/r05_enc_planes_tiled_ab.txt
