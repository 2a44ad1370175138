mkdir -p gpurun_out/r05j
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_compact.py tests/test_gpu_bf16x6.py tests/test_gpu_rgnn.py -q -x > gpurun_out/r05j/k.log 2>&1; echo "kernels rc=$?"; tail -4 gpurun_out/r05j/k.log | cut -c1-300
bash scripts/prof_step.sh r05j; head -c 300 gpurun_out/r05j/prof_bench.json
