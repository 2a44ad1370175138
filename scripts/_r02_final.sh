cd /tmp && export TMPDIR=/tmp && R=$GRAFT_REPO_ROOT && O=$R/gpurun_out/r02z && mkdir -p $O
timeout 300 python3 $R/bench.py > $O/bench_line.json 2> /dev/null
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o b -- python3 $R/bench.py --no-cpu-baseline > $O/prof_bench.json 2> $O/err.txt
rm -f $O/prof/*trace*
python3 - <<PY
import csv, json
rows=list(csv.DictReader(open("$O/prof/b_kernel_stats.csv")))
for r in rows:
    if "seg_sum_vec" in r["Name"]: print(r["Name"][:110], r["Calls"], float(r["AverageNs"])/1e3)
d=json.load(open("$O/bench_line.json")); print(d["value"], d["ms_per_step"], d["roofline"]["avg_us"], d["roofline"]["achieved"], d["cpu_baseline"]["value"])
PY
cd $R && timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -1
