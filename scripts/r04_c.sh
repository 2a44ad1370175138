#!/bin/bash
# round 4, call c: the register-accumulator endpoint sums (tests + micro-benchmark), composite test, 2-rank bench test
mkdir -p gpurun_out/r04c
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "one_pass or incidence" > gpurun_out/r04c/kernels.log 2>&1
echo "rc=$?" >> gpurun_out/r04c/kernels.log
timeout 300 python scripts/kbench_segacc.py > gpurun_out/r04c/kbench_segacc.json 2> gpurun_out/r04c/kbench_segacc.err
H=64 timeout 300 python scripts/kbench_segacc.py > gpurun_out/r04c/kbench_segacc64.json 2>> gpurun_out/r04c/kbench_segacc.err
timeout 1500 python -m pytest tests/test_gpu_bench_composite.py -x -q -m gpu > gpurun_out/r04c/composite.log 2>&1
echo "rc=$?" >> gpurun_out/r04c/composite.log
timeout 1500 python -m pytest tests/test_gpu_bench_line.py -x -q -m gpu > gpurun_out/r04c/benchline.log 2>&1
echo "rc=$?" >> gpurun_out/r04c/benchline.log
timeout 600 python bench.py --no-cpu-baseline > gpurun_out/r04c/bench.json 2> gpurun_out/r04c/bench.err
tail -3 gpurun_out/r04c/kernels.log; cat gpurun_out/r04c/kbench_segacc.json; tail -5 gpurun_out/r04c/composite.log; tail -5 gpurun_out/r04c/benchline.log
