#!/bin/bash
O=gpurun_out/r06p3; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && R=$GRAFT_REPO_ROOT
DMP_DEV_PREFETCH_LATE=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/p -o t -- python3 $R/bench.py --graph --no-cpu-baseline --no-all-outputs --no-gate-compact --no-gate-dense --extended-steps 0 > $R/$O/line.json 2> $R/$O/err.txt
python3 $R/scripts/step_timeline.py $R/$O/p/t_kernel_trace.csv > $R/$O/timeline_late.txt 2>&1
rm -f $R/$O/p/*trace*
head -75 $R/$O/timeline_late.txt
cd $R
timeout 1500 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_bench_composite.py tests/test_gpu_dmplayer.py tests/test_gpu_fullmodel.py tests/test_gpu_bf16x6.py tests/test_gpu_node_rows.py tests/test_gpu_layer0.py -q -m gpu -x > $O/t.log 2>&1
tail -8 $O/t.log | cut -c1-400
DMP_POISON_DEAD_ROWS=1 timeout 900 python -m pytest tests/test_gpu_bench_composite.py tests/test_gpu_dmplayer.py tests/test_gpu_fullmodel.py tests/test_gpu_node_rows.py -q -m gpu -x > $O/tp.log 2>&1
tail -5 $O/tp.log | cut -c1-400
