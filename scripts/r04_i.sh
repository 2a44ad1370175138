#!/bin/bash
mkdir -p gpurun_out/r04i
timeout 1800 python -m pytest tests/test_gpu_compact.py tests/test_gpu_graph.py tests/test_gpu_dp.py tests/test_gpu_harness.py tests/test_gpu_unc_harness.py tests/test_gpu_pins_r2.py tests/test_gpu_bench_line.py -x -q -m gpu > gpurun_out/r04i/tests.log 2>&1
echo "rc=$?" >> gpurun_out/r04i/tests.log
tail -25 gpurun_out/r04i/tests.log
