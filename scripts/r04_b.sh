#!/bin/bash
# round 4, call b: the one-pass endpoint sums (tests + micro-benchmark), composite test with diagnostics
mkdir -p gpurun_out/r04b
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "one_pass or incidence" > gpurun_out/r04b/kernels.log 2>&1
echo "rc=$?" >> gpurun_out/r04b/kernels.log
timeout 300 python scripts/kbench_segacc.py > gpurun_out/r04b/kbench_segacc.json 2> gpurun_out/r04b/kbench_segacc.err
H=64 timeout 300 python scripts/kbench_segacc.py > gpurun_out/r04b/kbench_segacc64.json 2>> gpurun_out/r04b/kbench_segacc.err
timeout 1500 python -m pytest tests/test_gpu_bench_composite.py -x -q -m gpu > gpurun_out/r04b/composite.log 2>&1
echo "rc=$?" >> gpurun_out/r04b/composite.log
tail -3 gpurun_out/r04b/kernels.log; cat gpurun_out/r04b/kbench_segacc.json; tail -5 gpurun_out/r04b/composite.log
