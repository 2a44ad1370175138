R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03d; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_dmplayer.py tests/test_gpu_rgnn.py -q -m gpu -x > $O/pytest.log 2>&1; tail -5 $O/pytest.log
timeout 300 python3 bench.py --no-cpu-baseline > $O/bench_x6.json 2> $O/err1.txt
DMP_EXACT_FP32=1 timeout 300 python3 bench.py --no-cpu-baseline > $O/bench_exact.json 2> $O/err2.txt
python3 - <<PY
import json
for n in ("x6","exact"):
    d=json.load(open("$O/bench_%s.json"%n))
    print(n, d["value"], d["ms_per_step"], d["step_ms_median"], {k.split("[")[0]:v["avg_us"] for k,v in d["kernels"].items() if ("typed" in k or "atb" in k or "mfma" in k) and v["avg_us"]>60})
PY
