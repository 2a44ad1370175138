#!/bin/bash
# round 4, call a: box facts, the composite parity test, today's default bench line
mkdir -p gpurun_out/r04a
(nproc; free -g; rocm-smi --showmeminfo vram | head -20) > gpurun_out/r04a/box.txt 2>&1
timeout 1500 python -m pytest tests/test_gpu_bench_composite.py -x -q -m gpu > gpurun_out/r04a/composite.log 2>&1
echo "composite rc=$?" >> gpurun_out/r04a/composite.log
timeout 600 python bench.py > gpurun_out/r04a/bench.json 2> gpurun_out/r04a/bench.err
echo "bench rc=$?" >> gpurun_out/r04a/bench.err
tail -5 gpurun_out/r04a/composite.log
