R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03o; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_layer0.py -x -q > $O/pytest.log 2>&1; tail -15 $O/pytest.log
timeout 300 python3 bench.py --no-cpu-baseline > $O/bench_l0.json 2> $O/err1.txt; tail -3 $O/err1.txt
DMP_LAYER0_NODES=0 timeout 300 python3 bench.py --no-cpu-baseline > $O/bench_l0n.json 2> $O/err3.txt
DMP_LAYER0=0 timeout 300 python3 bench.py --no-cpu-baseline > $O/bench_l0off.json 2> $O/err2.txt
python3 - <<PY
import json
for n in ("bench_l0","bench_l0n","bench_l0off"):
    try:
        d=json.load(open("$O/%s.json"%n)); print(n, d["value"], d["ms_per_step"], d["step_ms_median"], d["roofline"]["frac"])
    except Exception as e: print(n, "failed", e)
PY
bash scripts/_r03_q.sh > $O/prof.txt 2>&1; head -40 $O/prof.txt
