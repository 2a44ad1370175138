R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03m; mkdir -p $O; cd $R
for i in 1 2; do
DMP_EDGE_CHAIN=1 timeout 300 python3 bench.py --no-cpu-baseline > $O/chain1_$i.json 2>/dev/null
DMP_EDGE_CHAIN=0 timeout 300 python3 bench.py --no-cpu-baseline > $O/chain0_$i.json 2>/dev/null
done
python3 - <<PY
import json
for n in ("chain1_1","chain0_1","chain1_2","chain0_2"):
    d=json.load(open("$O/%s.json"%n)); print(n, d["value"], d["ms_per_step"], d["step_ms_median"], d["roofline"]["avg_us"], d["roofline"]["frac"], d["roofline_bwd"]["avg_us"], d["roofline_bwd"]["frac"], d["roofline"].get("avg_us_rocprof"), d["roofline"].get("traffic"))
PY
HID=64 ROUNDS=2 timeout 200 python3 scripts/host_time.py 2>&1 | tail -2; HID=128 ROUNDS=2 timeout 200 python3 scripts/host_time.py 2>&1 | tail -2
timeout 300 python3 bench.py --hid 64 --graph --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('h64 graph', d['value'], d['ms_per_step'])"
