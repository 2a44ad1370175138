cd /tmp && export TMPDIR=/tmp && R=$GRAFT_REPO_ROOT && O=$R/gpurun_out/r03j && mkdir -p $O
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o u -- python3 $R/scripts/kbench_unc.py > $O/unc.txt 2> $O/err.txt
rm -f $O/prof/*trace*
cat $O/unc.txt
