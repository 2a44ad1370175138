cd /tmp && export TMPDIR=/tmp && R=$GRAFT_REPO_ROOT && O=$R/gpurun_out/r03y && mkdir -p $O
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/prof -o u -- python3 $R/scripts/kbench_unc.py > $O/out.txt 2> $O/err.txt
python3 - <<PY
import csv
rows=list(csv.DictReader(open("$O/prof/u_kernel_trace.csv")))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
idx=[i for i,r in enumerate(rows) if "adamw_kernel" in r["Kernel_Name"]]
a,b=idx[-3],idx[-2]
t0=int(rows[a]["End_Timestamp"])
out=open("$O/step_sequence.txt","w")
prev=t0
for r in rows[a+1:b+1]:
    s,e=int(r["Start_Timestamp"]),int(r["End_Timestamp"])
    n=r["Kernel_Name"].replace("void dmp::(anonymous namespace)::","").replace("dmp::(anonymous namespace)::","").replace("void at::native::","")[:90]
    out.write("gap %5.1f dur %6.1f  %s\n"%((s-prev)/1e3,(e-s)/1e3,n))
    prev=e
out.close()
print("kernels in step", b-a, "span us", (int(rows[b]["End_Timestamp"])-t0)/1e3)
PY
rm -f $O/prof/u_kernel_trace.csv
cat $O/out.txt
