#!/bin/bash
# gpurun_out/<tag>z (scripts/gpu_evidence.sh) -> profiles/<tag>_* : the JSON lines (last line of each file), the kernel statistics,
# the PMC table and the micro-benchmarks.
TAG=${1:-r06}; O=gpurun_out/${TAG}z; P=profiles
last() { grep '^{' "$1" | tail -1; }
for n in bench_line bench_gate_dense_line config4_bench_line gloo_2rank_single_device_line gloo_2rank_config4_line bench_eager_line h64_bench_line rccl_one_rank_line rccl_one_rank_eager_line; do
  [ -s $O/$n.json ] && last $O/$n.json > $P/${TAG}_$n.json
done
[ -s $O/prof_bench_gate_compact.json ] && last $O/prof_bench_gate_compact.json > $P/${TAG}_bench_gate_compact_profiled_line.json
[ -s $O/prof_bench.json ] && last $O/prof_bench.json > $P/${TAG}_bench_profiled_line.json
cp $O/prof/b_kernel_stats.csv $P/${TAG}_bench_kernel_stats.csv
cp $O/profgc/g_kernel_stats.csv $P/${TAG}_bench_gate_compact_kernel_stats.csv
[ -s $O/profgd/d_kernel_stats.csv ] && cp $O/profgd/d_kernel_stats.csv $P/${TAG}_bench_gate_dense_kernel_stats.csv
[ -s $O/prof_bench_gate_dense.json ] && last $O/prof_bench_gate_dense.json > $P/${TAG}_bench_gate_dense_profiled_line.json
for n in pmc_h128.json profile_meta.json kbench_segacc.json kbench_atb.json bf16x6_probe.json unc.txt train_ragged.txt; do [ -s $O/$n ] && cp $O/$n $P/${TAG}_$n; done
ls -la $P | grep ${TAG}_
