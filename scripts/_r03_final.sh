# round 3 final evidence: bench line, rocprofv3 kernel stats, PMC traffic (separate passes), config 4, 2-rank plumbing run, UNC
cd /tmp && export TMPDIR=/tmp && R=$GRAFT_REPO_ROOT && O=$R/gpurun_out/r03z && mkdir -p $O
timeout 400 python3 $R/bench.py > $O/bench_line.json 2> $O/bench_err.txt
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o b -- python3 $R/bench.py --graph --no-cpu-baseline > $O/prof_bench.json 2> $O/err.txt
rm -f $O/prof/*trace*
timeout 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/f -o f -- python3 $R/bench.py --eager --steps 3 --warmup 2 --no-cpu-baseline > /dev/null 2> $O/err_f.txt
timeout 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/w -o w -- python3 $R/bench.py --eager --steps 3 --warmup 2 --no-cpu-baseline > /dev/null 2> $O/err_w.txt
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/k -o k -- python3 $R/bench.py --eager --steps 3 --warmup 2 --no-cpu-baseline > /dev/null 2> $O/err_k.txt
rm -f $O/k/*trace*
python3 - <<PY
import csv, collections, json
def load(path, counter):
    acc=collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r.get("Counter_Name")==counter:
            acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return acc
f=load("$O/f/f_counter_collection.csv","FETCH_SIZE"); w=load("$O/w/w_counter_collection.csv","WRITE_SIZE")
t={r["Name"]:float(r["AverageNs"])/1e3 for r in csv.DictReader(open("$O/k/k_kernel_stats.csv"))}
out={}
for k in f:
    if any(x in k for x in ("mfma_typed","mfma_pp","atb_k","atb_jobs","edge_chain_k","pool_relu_bwd_k","seg_sum_vec<32, true, false, true")):
        fv=[v for v in f[k] if v>0.5*max(f[k])]; wv=[v for v in w.get(k,[0]) if v>0.5*max(w.get(k,[1]))]
        hbm=(2*sum(fv)/len(fv)+ (sum(wv)/len(wv) if wv else 0))*1024
        short=k.replace("void dmp::(anonymous namespace)::","").split("(")[0]
        out[short]={"FETCH_SIZE_KB_avg_large":sum(fv)/len(fv),"WRITE_SIZE_KB_avg_large":(sum(wv)/len(wv) if wv else 0),"hbm_bytes_per_launch":hbm,"rocprof_avg_us_all_launches":t.get(k)}
        print("%-44s HBM %7.1f MB per large launch  (avg over all launches %6.1f us)" % (short, hbm/1e6, t.get(k,0)))
json.dump({"kernels":out,"correction":"hbm = (2*FETCH_SIZE + WRITE_SIZE)*1024 (FETCH_SIZE counts half of a 16-B/lane stream on gfx950, MI355X_MICROARCH.md); large = dispatches above half of the kernel's largest (the E-row launches)","command":"rocprofv3 --pmc FETCH_SIZE | --pmc WRITE_SIZE | --kernel-trace --stats (separate passes) -- python3 bench.py --eager --steps 3 --warmup 2 --no-cpu-baseline","shape":{"rows":73728,"edges":548864,"H":128}}, open("$O/pmc_h128.json","w"), indent=1)
PY
cd $R
timeout 600 python3 bench.py --workload 4 --steps 5 --warmup 2 --no-cpu-baseline > $O/config4_bench_line.json 2> $O/err_c4.txt
timeout 400 python3 bench.py --gpus 2 --backend gloo --single-device --steps 10 --warmup 3 --no-cpu-baseline 2> $O/err_2rank.txt | grep '^{' > $O/gloo_2rank_single_device_line.json
timeout 400 python3 bench.py --eager --no-cpu-baseline > $O/bench_eager_line.json 2> $O/err_eager.txt
timeout 400 python3 bench.py --hid 64 --no-cpu-baseline > $O/h64_bench_line.json 2> $O/err_h64.txt
timeout 300 python3 scripts/kbench_unc.py > $O/unc.txt 2>&1
python3 - <<PY
import json
for n in ("bench_line","config4_bench_line","gloo_2rank_single_device_line","bench_eager_line","h64_bench_line"):
    try:
        d=json.load(open("$O/%s.json"%n)); print(n, d["value"], d["ms_per_step"], d.get("n_gpus"), d.get("roofline") and d["roofline"]["frac"], d.get("roofline_bwd") and d["roofline_bwd"]["frac"])
    except Exception as e: print(n, "failed", e)
PY
cat $O/unc.txt
