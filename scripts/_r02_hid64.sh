cd /tmp && export TMPDIR=/tmp && R=$GRAFT_REPO_ROOT && O=$R/gpurun_out/r02h && mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o h -- python3 $R/bench.py --hid 64 --steps 10 --warmup 3 --no-cpu-baseline > $O/prof_h64.json 2>> $O/err.txt
rm -f $O/prof/*trace*
python3 - <<PY
import csv
rows=list(csv.DictReader(open("$O/prof/h_kernel_stats.csv")))
tot=sum(int(r["TotalDurationNs"]) for r in rows); calls=sum(int(r["Calls"]) for r in rows)
print("total GPU ms per step", tot/13/1e6, "launches per step", calls/13)
for r in rows[:45]:
    print("%-90s %5d %9.1f us  %5.2f%%" % (r["Name"][:90], int(r["Calls"]), float(r["AverageNs"])/1e3, float(r["Percentage"])))
PY
