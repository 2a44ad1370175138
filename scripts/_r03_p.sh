R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03p; mkdir -p $O; cd $R
timeout 300 python3 scripts/dbg_l0.py 2>&1 | tail -30
