set -x
mkdir -p gpurun_out
(timeout 2400 python -m pytest tests -m gpu -q -x --timeout=2000 --durations=15 > gpurun_out/r05_tests_all.log 2>&1; echo "rc=$?" >> gpurun_out/r05_tests_all.log)
tail -25 gpurun_out/r05_tests_all.log
python tools/bench_stream_kernels.py 2>/dev/null | grep '^{' > gpurun_out/r05_stream_kernels_after.jsonl
grep -i "epilogue\|sample_pl\|merge_sort" gpurun_out/r05_stream_kernels_after.jsonl
