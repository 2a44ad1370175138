mkdir -p gpurun_out/r05k
timeout 900 python -m pytest tests/test_gpu_node_rows.py tests/test_gpu_kernels.py tests/test_gpu_compact.py -q -x > gpurun_out/r05k/k.log 2>&1; echo "kernels rc=$?"; tail -8 gpurun_out/r05k/k.log | cut -c1-300
bash scripts/prof_step.sh r05k; head -c 300 gpurun_out/r05k/prof_bench.json
