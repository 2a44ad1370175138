#!/bin/bash
# rocprofv3 --kernel-trace --stats of the replayed default step -> gpurun_out/<tag>/prof/b_kernel_stats.csv (+ the line)
TAG=${1:-r05p}
cd /tmp && export TMPDIR=/tmp && R=$GRAFT_REPO_ROOT && O=$R/gpurun_out/${TAG} && mkdir -p $O
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o b -- python3 $R/bench.py --graph --no-cpu-baseline --no-all-outputs --no-gate-compact --no-gate-dense --extended-steps 0 > $O/prof_bench.json 2> $O/err.txt
rm -f $O/prof/*trace*
ls $O/prof | head
