#!/bin/bash
mkdir -p gpurun_out/r04d
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "one_pass or incidence" > gpurun_out/r04d/kernels.log 2>&1
echo "rc=$?" >> gpurun_out/r04d/kernels.log
timeout 300 python scripts/kbench_segacc.py > gpurun_out/r04d/kbench_segacc.json 2> gpurun_out/r04d/kbench_segacc.err
H=64 timeout 300 python scripts/kbench_segacc.py > gpurun_out/r04d/kbench_segacc64.json 2>> gpurun_out/r04d/kbench_segacc.err
tail -3 gpurun_out/r04d/kernels.log; cat gpurun_out/r04d/kbench_segacc.json gpurun_out/r04d/kbench_segacc64.json
