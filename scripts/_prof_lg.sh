cd /tmp && export TMPDIR=/tmp && R=$GRAFT_REPO_ROOT && O=$R/gpurun_out/lg && mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o lg -- python3 $R/scripts/prof_linegraph.py 10 > $O/out.txt 2> $O/err.txt
rm -f $O/prof/*trace*
python3 - <<PY
import csv
rows=list(csv.DictReader(open("$O/prof/lg_kernel_stats.csv")))
tot=sum(int(r["TotalDurationNs"]) for r in rows)
print("total kernel ms per transform %.3f, launches %.0f" % (tot/10/1e6, sum(int(r["Calls"]) for r in rows)/10))
for r in rows[:22]:
    print("%-100s %5d %9.1f us %5.1f%%" % (r["Name"].replace("void ","")[:100], int(r["Calls"]), float(r["AverageNs"])/1e3, float(r["Percentage"])))
PY
