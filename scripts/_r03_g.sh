cd /tmp && export TMPDIR=/tmp && R=$GRAFT_REPO_ROOT && O=$R/gpurun_out/r03g && mkdir -p $O
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o b -- python3 $R/bench.py --no-cpu-baseline > $O/prof_bench.json 2> $O/err.txt
rm -f $O/prof/*trace*
ls $O/prof
cd $R && timeout 900 python -m pytest tests/test_gpu_shapes.py -q -m gpu -x 2>&1 | tail -3
