R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02g; mkdir -p $O; cd $R
python bench.py --graph --no-cpu-baseline > $O/bench_graph_line.json 2>/dev/null
python bench.py --hid 64 --graph --no-cpu-baseline > $O/bench_h64_graph_line.json 2>/dev/null
python bench.py --hid 64 --no-cpu-baseline > $O/bench_h64_line.json 2>/dev/null
python bench.py --no-cpu-baseline > $O/bench_line.json 2>/dev/null
(python scripts/kbench_train_small.py; B=64 python scripts/kbench_train_small.py) 2>&1 | grep -v amdgpu > $O/train_small.txt
for f in bench_graph_line bench_h64_graph_line bench_h64_line bench_line; do python -c "
import json; d=json.load(open('$O/$f.json')); print('$f', d['value'], d['ms_per_step'], d['config'].get('launch'))"; done; cat $O/train_small.txt
