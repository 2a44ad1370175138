R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03i; mkdir -p $O; cd $R
timeout 1200 python -m pytest tests/test_gpu_shapes.py tests/test_gpu_kernels.py tests/test_gpu_rgnn.py tests/test_gpu_dp.py -q -m gpu -x > $O/pytest.log 2>&1; tail -4 $O/pytest.log
timeout 600 python3 bench.py --workload 4 --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_c4_m1.json 2> $O/err_c4.txt; tail -3 $O/err_c4.txt
timeout 300 python3 bench.py --no-cpu-baseline > $O/bench_c2.json 2> $O/err_c2.txt
python3 - <<PY
import json
for n in ("c4_m1","c2"):
    try:
        d=json.load(open("$O/bench_%s.json"%n))
        print(n, d["value"], d["ms_per_step"], d["config"]["micro_batches"], d["config"]["peak_hbm_allocated_gb"], d["roofline"] and d["roofline"]["frac"], d["roofline_bwd"] and d["roofline_bwd"]["frac"])
    except Exception as e: print(n, "failed", e)
PY
