#!/bin/bash
# dev: tr_b16 probe + workgroups-per-CU probe of the typed kernels + the targeted tests of this round's host-side changes
O=gpurun_out/r06p1; mkdir -p $O
hipcc --offload-arch=gfx950 -O3 -o /tmp/tr scripts/tr_probe.hip 2>/dev/null && /tmp/tr > $O/tr.txt 2>&1
bash scripts/quick_bench.sh r06p1q "DMP_DEV_TYPED_PER_CU=1" > $O/q.txt 2>&1
cp gpurun_out/r06p1q/stats.txt $O/stats_default.txt
cd /tmp && export TMPDIR=/tmp && R=$GRAFT_REPO_ROOT
DMP_DEV_TYPED_PER_CU=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/p1 -o t -- python3 $R/bench.py --graph --no-cpu-baseline --no-all-outputs --no-gate-compact --no-gate-dense --extended-steps 0 > $R/$O/prof1_line.json 2> $R/$O/prof1_err.txt
python3 $R/scripts/show_stats.py $R/$O/p1/t_kernel_stats.csv > $R/$O/stats_percu1.txt 2>&1
rm -f $R/$O/p1/*trace*
cd $R
timeout 1400 python -m pytest tests/test_gpu_build_current.py tests/test_gpu_fullmodel.py tests/test_gpu_dmplayer.py tests/test_gpu_side_stream.py tests/test_gpu_ragged_replay.py tests/test_gpu_compact.py tests/test_gpu_dp.py tests/test_gpu_graph.py tests/test_gpu_harness.py -q -m gpu > $O/t.log 2>&1
tail -25 $O/t.log | cut -c1-400
cat $O/q.txt
