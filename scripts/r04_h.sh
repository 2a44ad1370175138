#!/bin/bash
mkdir -p gpurun_out/r04h; rm -f gpurun_out/r04h/log6.txt
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "one_pass" 2>&1 | tail -2 >> gpurun_out/r04h/log6.txt
ITERS=1500 timeout 900 python3 scripts/stress_segacc.py 2>&1 | grep iterations >> gpurun_out/r04h/log6.txt
timeout 300 python scripts/kbench_segacc.py 2>/dev/null | cut -c1-520 >> gpurun_out/r04h/log6.txt
run() { echo "== $*" >> gpurun_out/r04h/log6.txt; timeout 300 python3 bench.py --gpus 2 --backend gloo --single-device --steps 8 --warmup 3 --no-cpu-baseline $1 2>&1 | grep -v "amdgpu.ids\|socket.cpp\|^\[rank\|Gloo" | cut -c1-160 >> gpurun_out/r04h/log6.txt; echo "rc=${PIPESTATUS[0]}" >> gpurun_out/r04h/log6.txt; }
run "--eager"; run ""; run ""; run "--batch 512"
cat gpurun_out/r04h/log6.txt | cut -c1-520
