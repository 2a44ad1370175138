#!/bin/bash
# Dev aid: quick bench lines (no CPU baseline, no controls) for a list of environment settings, + kernel stats of the first.
#   usage: quick_bench.sh <tag> ["ENV=.. ENV2=.." ...]
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp && R=$GRAFT_REPO_ROOT && O=$R/gpurun_out/$TAG && mkdir -p $O
Q="--graph --no-cpu-baseline --no-all-outputs --no-gate-compact --no-gate-dense --extended-steps 100 $QB_EXTRA"   # QB_EXTRA: e.g. "--filter-net None"
i=0
for e in "" "$@"; do
  env $e timeout 300 python3 $R/bench.py $Q > $O/line_$i.json 2> $O/err_$i.txt
  python3 - <<PY
import json
try:
    d = json.loads([l for l in open("$O/line_$i.json") if l.startswith("{")][-1])
    x = d.get("steps_extended") or {}
    print("[$e]", d["ms_per_step"], "ext", x.get("ms_per_step"), "fwd us", d["roofline"]["avg_us"], d["roofline"]["frac"], "bwd us", d["roofline_bwd"]["avg_us"], d["roofline_bwd"]["frac"])
except Exception as ex:
    print("[$e] failed", ex)
PY
  i=$((i+1))
done
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/p -o t -- python3 $R/bench.py --graph --no-cpu-baseline --no-all-outputs --no-gate-compact --no-gate-dense --extended-steps 0 $QB_EXTRA > $O/prof_line.json 2> $O/prof_err.txt
python3 $R/scripts/step_timeline.py $O/p/t_kernel_trace.csv > $O/timeline.txt 2>&1
python3 $R/scripts/show_stats.py $O/p/t_kernel_stats.csv > $O/stats.txt 2>&1
rm -f $O/p/*trace*
tail -1 $O/timeline.txt; head -1 $O/stats.txt
