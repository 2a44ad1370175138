R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03t; mkdir -p $O; cd $R
for i in 1 2; do
timeout 300 python3 bench.py --eager --no-cpu-baseline > $O/e$i.json 2> $O/err_e$i.txt
timeout 300 python3 bench.py --no-cpu-baseline > $O/g$i.json 2> $O/err_g$i.txt
done
python3 - <<PY
import json
for n in ("e1","g1","e2","g2"):
    try:
        d=json.load(open("$O/%s.json"%n)); print(n, d["value"], d["ms_per_step"], d["step_ms_min"], d["step_ms_median"], d["step_ms_max"], d["roofline"]["frac"], d["roofline"]["launches"], d["config"]["launch"][:30], d["config"].get("eager_ms_per_step"))
    except Exception as e: print(n, "failed", e)
PY
tail -3 $O/err_g1.txt
