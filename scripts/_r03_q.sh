cd /tmp && export TMPDIR=/tmp && R=$GRAFT_REPO_ROOT && O=$R/gpurun_out/r03q && mkdir -p $O
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o b -- python3 $R/bench.py --no-cpu-baseline > $O/prof_bench.json 2> $O/err.txt
rm -f $O/prof/*trace*
python3 - <<PY
import csv
rows=list(csv.DictReader(open("$O/prof/b_kernel_stats.csv")))
tot=sum(float(r['TotalDurationNs']) for r in rows)
for r in sorted(rows,key=lambda r:-float(r['TotalDurationNs']))[:48]:
    n=r['Name'].replace('void dmp::(anonymous namespace)::','').replace('dmp::(anonymous namespace)::','')[:70]
    print("%6.2f%% calls %5s avg %8.1f us  %s"%(100*float(r['TotalDurationNs'])/tot, r['Calls'], float(r['AverageNs'])/1e3, n))
PY
