R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03v; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_batchnorm.py tests/test_gpu_unc_harness.py tests/test_gpu_unc_sampling.py -x -q > $O/pytest.log 2>&1; tail -15 $O/pytest.log
timeout 300 python3 scripts/kbench_unc.py 2>&1 | tail -3
DMP_HIP_BATCHNORM=0 timeout 300 python3 scripts/kbench_unc.py 2>&1 | tail -2
