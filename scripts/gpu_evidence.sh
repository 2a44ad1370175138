#!/bin/bash
# Evidence of a round, taken on the GPU box in one call (gpurun -- 'bash scripts/gpu_evidence.sh r05'):
#   <tag>z/bench_line.json            the default bench line (with the CPU baseline)
#   <tag>z/prof/b_kernel_stats.csv    rocprofv3 --kernel-trace --stats of the replayed default mode
#   <tag>z/pmc_h128.json              HBM traffic per launch of the E-row kernels: FETCH_SIZE and WRITE_SIZE in SEPARATE --pmc passes
#                                     (gpurun refuses counters together with traces), FETCH_SIZE doubled (MI355X_MICROARCH.md, HBM)
#   <tag>z/profile_meta.json          launch shape + content hash of the kernel sources these profiles were taken with
#   config 4 shard, eager / hid 64 lines, 2-rank plumbing line, UNC, micro-benchmarks
# Copy what is to be judged into profiles/ (tracked) afterwards.
TAG=${1:-r06}
cd /tmp && export TMPDIR=/tmp && R=$GRAFT_REPO_ROOT && O=$R/gpurun_out/${TAG}z && mkdir -p $O
timeout 900 python3 $R/bench.py --cpu-b1024 > $O/bench_line.json 2> $O/bench_err.txt
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o b -- python3 $R/bench.py --graph --no-cpu-baseline --no-all-outputs --no-gate-compact --no-gate-dense --extended-steps 0 > $O/prof_bench.json 2> $O/err.txt
rm -f $O/prof/*trace*
timeout 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/f -o f -- python3 $R/bench.py --eager --steps 3 --warmup 2 --no-cpu-baseline --no-all-outputs --no-gate-compact --no-gate-dense --extended-steps 0 > /dev/null 2> $O/err_f.txt
timeout 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/w -o w -- python3 $R/bench.py --eager --steps 3 --warmup 2 --no-cpu-baseline --no-all-outputs --no-gate-compact --no-gate-dense --extended-steps 0 > /dev/null 2> $O/err_w.txt
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/k -o k -- python3 $R/bench.py --eager --steps 3 --warmup 2 --no-cpu-baseline --no-all-outputs --no-gate-compact --no-gate-dense --extended-steps 0 > /dev/null 2> $O/err_k.txt
rm -f $O/k/*trace*
# the ALL-ROWS step (no filter net: `gate_dense`) as the headline of its own line, and its kernel statistics
timeout 300 python3 $R/bench.py --graph --filter-net None --no-cpu-baseline --no-all-outputs --no-gate-compact --extended-steps 50 > $O/bench_gate_dense_line.json 2> $O/err_gd_line.txt
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/profgd -o d -- python3 $R/bench.py --graph --filter-net None --no-cpu-baseline --no-all-outputs --no-gate-compact --extended-steps 0 > $O/prof_bench_gate_dense.json 2> $O/err_gd.txt
rm -f $O/profgd/*trace*
# the gate-compact mode of the same step (bench.py's `gate_compact` object): its own kernel statistics
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/profgc -o g -- python3 $R/bench.py --graph --gate-compact --no-cpu-baseline --no-all-outputs --extended-steps 0 > $O/prof_bench_gate_compact.json 2> $O/err_gc.txt
rm -f $O/profgc/*trace*
cd $R
python3 - <<PY
import csv, collections, json, sys
sys.path.insert(0, "$R")
from dualmessagepassing_amd import _build
def load(path, counter):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r.get("Counter_Name") == counter:
            acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return acc
f = load("$O/f/f_counter_collection.csv", "FETCH_SIZE"); w = load("$O/w/w_counter_collection.csv", "WRITE_SIZE")
t = {r["Name"]: float(r["AverageNs"]) / 1e3 for r in csv.DictReader(open("$O/k/k_kernel_stats.csv"))}
out = {}
for k in f:
    if any(x in k for x in ("mfma_typed", "mfma_pp", "atb_k", "atb_jobs", "pool_relu_bwd_k", "seg_sum_vec<32, true, true, true", "seg_sum_vec<32, true, false, true", "seg_acc_graphs_k", "l0_edge_fwd_k", "l0_bwd_w_k", "h1w_k", "dzw_k", "atb2_k")):
        fv = [v for v in f[k] if v > 0.5 * max(f[k])]; wv = [v for v in w.get(k, [0]) if v > 0.5 * max(w.get(k, [1]))]
        hbm = (2 * sum(fv) / len(fv) + (sum(wv) / len(wv) if wv else 0)) * 1024
        short = k.replace("void ", "").replace("dmp::(anonymous namespace)::", "").split("(")[0]
        out[short] = {"FETCH_SIZE_KB_avg_large": sum(fv) / len(fv), "WRITE_SIZE_KB_avg_large": (sum(wv) / len(wv) if wv else 0),
                      "hbm_bytes_per_launch": hbm, "rocprof_avg_us_all_launches": t.get(k)}
        print("%-44s HBM %7.1f MB per large launch  (avg over all launches %6.1f us)" % (short, hbm / 1e6, t.get(k, 0)))
json.dump({"kernels": out, "correction": "hbm = (2*FETCH_SIZE + WRITE_SIZE)*1024 (FETCH_SIZE counts half of a 16-B/lane stream on gfx950, MI355X_MICROARCH.md; the one-pass endpoint sums load 4 B per lane: the same doubling reproduces their byte count); large = dispatches above half of the kernel's largest (the E-row launches)",
           "command": "rocprofv3 --pmc FETCH_SIZE | --pmc WRITE_SIZE | --kernel-trace --stats (separate passes) -- python3 bench.py --eager --steps 3 --warmup 2 --no-cpu-baseline --no-all-outputs --no-gate-compact --no-gate-dense --extended-steps 0",
           "shape": {"rows": 73728, "edges": 548864, "H": 128}}, open("$O/pmc_h128.json", "w"), indent=1)
json.dump({"rows": 73728, "edges": 548864, "H": 128, "lib_srchash": _build.source_hash(),
           "command": "rocprofv3 --kernel-trace --stats -- python3 bench.py --graph --no-cpu-baseline --no-all-outputs --no-gate-compact --no-gate-dense --extended-steps 0 (kernel stats: the default mode of bench.py at N = 1, run in the profiled process itself); rocprofv3 --pmc FETCH_SIZE | --pmc WRITE_SIZE | --kernel-trace --stats -- python3 bench.py --eager --steps 3 --warmup 2 --no-cpu-baseline --no-all-outputs --no-gate-compact --no-gate-dense --extended-steps 0 (PMC, separate passes)",
           "note": "launch shape of the two scatter-add launches of bench.py (union of 1024 pattern + target graphs, hid 128); lib_srchash = content hash of the kernel sources (dualmessagepassing_amd/_build.py::source_hash): bench.py quotes these profiles only for the same build"},
          open("$O/profile_meta.json", "w"), indent=1)
PY
timeout 600 python3 bench.py --workload 4 --steps 5 --warmup 2 --no-cpu-baseline > $O/config4_bench_line.json 2> $O/err_c4.txt
timeout 500 python3 bench.py --gpus 2 --backend gloo --single-device --steps 10 --warmup 3 --no-cpu-baseline 2> $O/err_2rank.txt | grep '^{' > $O/gloo_2rank_single_device_line.json
timeout 500 python3 bench.py --gpus 2 --backend gloo --single-device --workload 4 --batch 256 --steps 5 --warmup 3 --no-cpu-baseline 2> $O/err_2rank_c4.txt | grep '^{' > $O/gloo_2rank_config4_line.json
timeout 400 python3 bench.py --eager --no-cpu-baseline > $O/bench_eager_line.json 2> $O/err_eager.txt
timeout 400 python3 bench.py --hid 64 --no-cpu-baseline > $O/h64_bench_line.json 2> $O/err_h64.txt
timeout 300 python3 scripts/kbench_unc.py > $O/unc.txt 2>&1
UNC_HID=50 timeout 300 python3 scripts/kbench_unc.py >> $O/unc.txt 2>&1
timeout 300 python3 scripts/kbench_segacc.py > $O/kbench_segacc.json 2> $O/kbench_segacc.err
timeout 400 python3 bench.py --force-collective --no-cpu-baseline --extended-steps 0 > $O/rccl_one_rank_line.json 2> $O/err_rccl.txt
timeout 400 python3 bench.py --force-collective --eager --no-cpu-baseline --extended-steps 0 > $O/rccl_one_rank_eager_line.json 2> $O/err_rccl_e.txt
timeout 600 python3 scripts/kbench_train_ragged.py > $O/train_ragged.txt 2>&1
timeout 300 python3 scripts/kbench_atb.py > $O/kbench_atb.json 2> $O/kbench_atb.err
timeout 300 python3 scripts/bf16x6_probe.py > $O/bf16x6_probe.json 2> $O/bf16x6.err
python3 - <<PY
import json
for n in ("bench_line", "config4_bench_line", "gloo_2rank_single_device_line", "gloo_2rank_config4_line", "bench_eager_line", "h64_bench_line"):
    try:
        d = json.loads([l for l in open("$O/%s.json" % n) if l.startswith("{")][-1]); print(n, d["value"], d["ms_per_step"], d.get("n_gpus"), d.get("launch_mode"), d.get("roofline") and d["roofline"]["frac"], d.get("roofline_bwd") and d["roofline_bwd"]["frac"], d.get("all_outputs_ms_per_step"))
    except Exception as e: print(n, "failed", e)
PY
cat $O/unc.txt
rm -rf $O/f $O/w
