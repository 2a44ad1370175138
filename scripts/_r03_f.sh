R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03f; mkdir -p $O; cd $R
timeout 1800 python -m pytest tests -q -m gpu > $O/pytest.log 2>&1; tail -5 $O/pytest.log
