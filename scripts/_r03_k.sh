R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03k; mkdir -p $O; cd $R
timeout 300 python3 scripts/kbench_unc.py > $O/unc_before_tune.txt 2>&1; cat $O/unc_before_tune.txt | tail -2
DMP_TUNE_OUT=$O/tune_unc.csv timeout 900 python3 scripts/kbench_unc.py > $O/unc_tuning.txt 2>&1; tail -2 $O/unc_tuning.txt
wc -l $O/tune_unc.csv
