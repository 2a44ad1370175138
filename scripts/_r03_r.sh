cd /tmp && export TMPDIR=/tmp && R=$GRAFT_REPO_ROOT && O=$R/gpurun_out/r03r && mkdir -p $O
rm -f $R/scripts/_dbg/libl0_[1-9]*.so
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/f -o f -- python3 $R/scripts/mb_l0.py > /dev/null 2> $O/err_f.txt
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/w -o w -- python3 $R/scripts/mb_l0.py > /dev/null 2> $O/err_w.txt
python3 - <<PY
import csv, collections
for tag, name in (("f","FETCH_SIZE"),("w","WRITE_SIZE")):
    acc=collections.defaultdict(list)
    for r in csv.DictReader(open("$O/%s/%s_counter_collection.csv"%(tag,tag))):
        if r.get("Counter_Name")==name: acc[r["Kernel_Name"][:60]].append(float(r["Counter_Value"]))
    for k,v in acc.items():
        if "l0_" in k: print(name, k, "n=%d"%len(v), "avg KB %.0f"%(sum(v)/len(v)), "max %.0f"%max(v))
PY
