#!/bin/bash
O=gpurun_out/r06p7; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_compact.py -q -m gpu -k "one_launch or piece_image or identity or kept_csr or without_a_gate" > $O/t.log 2>&1
tail -6 $O/t.log | cut -c1-700
bash scripts/quick_bench.sh r06p7q > $O/q.txt 2>&1
cat $O/q.txt
grep -n "dzw_k\|atb2_k\|h1w_k\|reduce_partials\|mfma_typed<4" gpurun_out/r06p7q/timeline.txt
timeout 1200 python -m pytest tests/test_gpu_bench_composite.py tests/test_gpu_bench_line.py tests/test_gpu_fullmodel.py tests/test_gpu_dmplayer.py tests/test_gpu_lazy_rows.py -q -m gpu -x > $O/t2.log 2>&1
tail -6 $O/t2.log | cut -c1-500
