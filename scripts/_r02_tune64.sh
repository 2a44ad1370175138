R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02t; mkdir -p $O; cd $R
PYTORCH_TUNABLEOP_ENABLED=1 PYTORCH_TUNABLEOP_TUNING=1 PYTORCH_TUNABLEOP_FILENAME=$O/tune64.csv timeout 1500 python bench.py --hid 64 --steps 3 --warmup 2 --no-cpu-baseline > $O/tune_line.json 2> $O/tune_err.txt
ls -la $O; wc -l $O/tune64*.csv; tail -3 $O/tune_err.txt
