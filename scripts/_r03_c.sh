R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03c; mkdir -p $O; cd $R
timeout 600 python3 scripts/mb_typed_dbg.py > $O/typed_dbg.txt 2>&1; cat $O/typed_dbg.txt
