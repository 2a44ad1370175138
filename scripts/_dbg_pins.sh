mkdir -p gpurun_out/r05l
timeout 900 python -m pytest tests/test_gpu_node_rows.py tests/test_gpu_compact.py -q -x > gpurun_out/r05l/k.log 2>&1; echo "rc=$?"; tail -4 gpurun_out/r05l/k.log | cut -c1-300
bash scripts/prof_step.sh r05l; head -c 300 gpurun_out/r05l/prof_bench.json
