R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03b; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_gpu_fullmodel.py tests/test_gpu_graph.py tests/test_gpu_dmplayer.py tests/test_gpu_shapes.py -q -m gpu -k "config_1 or indexing or state_dict or fused_rep or config5 or full_config2" > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -30 $O/pytest.log
