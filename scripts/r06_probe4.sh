#!/bin/bash
O=gpurun_out/r06p4; mkdir -p $O
bash scripts/quick_bench.sh r06p4q "DMP_DEV_PREFETCH_LATE=1 DMP_DEV_FORK_EVENT=1" "DMP_DEV_SIDE_WARM=1" "DMP_DEV_SIDE_WARM=1 DMP_DEV_PREFETCH_LATE=1 DMP_DEV_FORK_EVENT=1" > $O/q.txt 2>&1
cat $O/q.txt
cd /tmp && export TMPDIR=/tmp && R=$GRAFT_REPO_ROOT
DMP_DEV_PREFETCH_LATE=1 DMP_DEV_FORK_EVENT=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/p -o t -- python3 $R/bench.py --graph --no-cpu-baseline --no-all-outputs --no-gate-compact --no-gate-dense --extended-steps 0 > $R/$O/line.json 2> $R/$O/err.txt
python3 $R/scripts/step_timeline.py $R/$O/p/t_kernel_trace.csv > $R/$O/timeline_ev.txt 2>&1
rm -f $R/$O/p/*trace*
head -70 $R/$O/timeline_ev.txt
