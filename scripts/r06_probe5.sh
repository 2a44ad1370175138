#!/bin/bash
O=gpurun_out/r06p5; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_gate_bundle.py -q -m gpu -k "piece_image or second_linear or bundle or weight_gradient or class" > $O/t.log 2>&1
tail -12 $O/t.log | cut -c1-700
bash scripts/quick_bench.sh r06p5q "DMP_DEV_PREFETCH_LATE=1" > $O/q.txt 2>&1
cat $O/q.txt
head -32 gpurun_out/r06p5q/stats.txt
