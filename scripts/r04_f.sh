#!/bin/bash
mkdir -p gpurun_out/r04f
timeout 1500 python -m pytest tests/test_gpu_fullmodel.py tests/test_gpu_harness.py tests/test_gpu_pins_r2.py tests/test_gpu_lazy_rows.py -q -m gpu > gpurun_out/r04f/tests.log 2>&1
echo "rc=$?" >> gpurun_out/r04f/tests.log
timeout 600 python scripts/trace_small_ops.py > gpurun_out/r04f/small_ops.txt 2>&1
tail -8 gpurun_out/r04f/tests.log
