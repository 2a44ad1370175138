R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03e; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_gpu_gemm6.py -q -m gpu -x > $O/pytest.log 2>&1; tail -15 $O/pytest.log
timeout 300 python3 scripts/kbench_gemm6.py > $O/kbench_gemm6.txt 2>&1; cat $O/kbench_gemm6.txt
