#!/bin/bash
O=gpurun_out/r06p6; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "piece_image" > $O/t.log 2>&1; timeout 600 python -m pytest tests/test_gpu_compact.py -q -m gpu -k "kept" >> $O/t.log 2>&1
tail -5 $O/t.log | cut -c1-700
bash scripts/quick_bench.sh r06p6q "DMP_DEV_ATB2=0" > $O/q.txt 2>&1
cat $O/q.txt
grep -n "atb2_k\|h1w_k\|reduce_partials\|atb_k\|atb_jobs" gpurun_out/r06p6q/timeline.txt
