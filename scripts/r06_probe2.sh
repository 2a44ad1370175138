#!/bin/bash
# dev: the one-launch second-Linear backward: its kernel test, then the step with / without it, then queue ids of a replayed step
O=gpurun_out/r06p2; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "second_linear" > $O/t.log 2>&1
tail -15 $O/t.log | cut -c1-600
bash scripts/quick_bench.sh r06p2q "DMP_DEV_PREFETCH_LATE=1" > $O/q.txt 2>&1
cat $O/q.txt
grep -n "bwd_h1_w\|h1w_k\|atb_k<1\|mfma_typed<32" gpurun_out/r06p2q/stats.txt
head -60 gpurun_out/r06p2q/timeline.txt
