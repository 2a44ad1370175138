cd /tmp && export TMPDIR=/tmp && R=$GRAFT_REPO_ROOT && O=$R/gpurun_out/r03u && mkdir -p $O
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o u -- python3 $R/scripts/kbench_unc.py > $O/out.txt 2> $O/err.txt
rm -f $O/prof/*trace*
python3 - <<PY
import csv
rows=list(csv.DictReader(open("$O/prof/u_kernel_stats.csv")))
tot=sum(float(r['TotalDurationNs']) for r in rows); calls=sum(int(r['Calls']) for r in rows)
print("kernels", len(rows), "calls", calls, "total ms", tot/1e6)
for r in sorted(rows,key=lambda r:-int(r['Calls']))[:45]:
    n=r['Name'].replace('void dmp::(anonymous namespace)::','').replace('dmp::(anonymous namespace)::','')[:110]
    print("calls %5s avg %7.1f us tot %6.2f%%  %s"%(r['Calls'], float(r['AverageNs'])/1e3, 100*float(r['TotalDurationNs'])/tot, n))
PY
cat $O/out.txt
