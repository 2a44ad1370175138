R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03w; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q 2>&1 | tail -2
for i in 1 2; do timeout 300 python3 bench.py --graph --no-cpu-baseline > $O/a$i.json 2> $O/err_a$i.txt; done
sed -i 's/#define DMP_SEG_VAR 1 /#define DMP_SEG_VAR 65 /' dualmessagepassing_amd/csrc/dmp_agg.hip
for i in 1 2; do timeout 600 python3 bench.py --graph --no-cpu-baseline > $O/b$i.json 2> $O/err_b$i.txt; done
python3 - <<PY
import json
for n in ("a1","a2","b1","b2"):
    try:
        d=json.load(open("$O/%s.json"%n)); print(n, d["value"], d["ms_per_step"], d["step_ms_median"], d["roofline"]["frac"], d["roofline"]["avg_us"], d["roofline_bwd"]["frac"], d["roofline_bwd"]["avg_us"])
    except Exception as e: print(n, "failed", e)
PY
