R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03i; mkdir -p $O; cd $R
timeout 1200 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_dmplayer.py tests/test_gpu_rgnn.py tests/test_gpu_shapes.py tests/test_gpu_fullmodel.py -q -m gpu -x > $O/pytest.log 2>&1; tail -6 $O/pytest.log
timeout 600 python3 bench.py --workload 4 --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_c4_m1.json 2> $O/err_c4.txt; tail -3 $O/err_c4.txt
timeout 600 python3 bench.py --workload 4 --steps 5 --warmup 2 --no-cpu-baseline --micro-batches 4 > $O/bench_c4_m4.json 2> $O/err_c4b.txt
timeout 300 python3 bench.py --no-cpu-baseline > $O/bench_c2.json 2> $O/err_c2.txt
python3 - <<PY
import json
for n in ("c4_m1","c4_m4","c2"):
    try:
        d=json.load(open("$O/bench_%s.json"%n))
        print(n, d["value"], d["ms_per_step"], d["config"]["micro_batches"], d["config"]["peak_hbm_allocated_gb"], d["roofline"] and d["roofline"]["frac"])
    except Exception as e: print(n, "failed", e)
PY
