R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03x; mkdir -p $O; cd $R
timeout 1200 python -m pytest tests/test_gpu_fullmodel.py tests/test_gpu_layer0.py tests/test_gpu_graph.py tests/test_gpu_pins_r2.py -x -q > $O/pytest.log 2>&1; tail -6 $O/pytest.log
for i in 1 2; do timeout 300 python3 bench.py --graph --no-cpu-baseline > $O/a$i.json 2> $O/err_a$i.txt; done
python3 - <<PY
import json
for n in ("a1","a2"):
    try:
        d=json.load(open("$O/%s.json"%n)); print(n, d["value"], d["ms_per_step"], d["step_ms_median"], d["roofline"]["frac"])
    except Exception as e: print(n, "failed", e)
PY
