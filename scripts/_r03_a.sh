# round 3, first GPU call: GPU suite, bench line (new fields), self-spawned 2-rank run on one device (gloo)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03a; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -15 $O/pytest.log
timeout 400 python3 bench.py > $O/bench_line.json 2> $O/bench_err.txt; echo "bench rc=$?"
timeout 400 python3 bench.py --gpus 2 --backend gloo --single-device --steps 10 --warmup 3 > $O/bench_2rank_gloo.json 2> $O/bench_2rank_err.txt; echo "2rank rc=$?"
python3 - <<PY
import json
d=json.load(open("$O/bench_line.json"))
print({k:d[k] for k in ("value","ms_per_step","step_ms_min","step_ms_median","step_ms_max")}, d["roofline"]["avg_us"], d["roofline"]["frac"], d["cpu_baseline"]["value"], d.get("cpu_baseline_b1024",{}).get("value"))
e=json.load(open("$O/bench_2rank_gloo.json")); print({k:e[k] for k in ("value","n_gpus","ranks_seen","backend","devices","launcher","ms_per_step")})
PY
