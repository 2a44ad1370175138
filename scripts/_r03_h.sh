R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03h; mkdir -p $O; cd $R
timeout 900 python -m pytest tests -q -m gpu -x -k "compgcn or CompGCN or layers_other or pins_r2 or harness" > $O/pytest.log 2>&1; tail -6 $O/pytest.log
timeout 300 python3 scripts/kbench_compgcn.py --json $O/compgcn.json > $O/kbench_compgcn.txt 2>&1; cat $O/kbench_compgcn.txt
