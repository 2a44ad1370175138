R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03l; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fullmodel.py tests/test_gpu_pins_r2.py -q -m gpu -x > $O/pytest.log 2>&1; tail -6 $O/pytest.log
timeout 300 python3 bench.py --no-cpu-baseline > $O/bench.json 2> $O/err1.txt; tail -2 $O/err1.txt
python3 - <<PY
import json
d=json.load(open("$O/bench.json"))
print(d["value"], d["ms_per_step"], d["step_ms_median"], {k.split("[")[0]:v["avg_us"] for k,v in d["kernels"].items() if v["avg_us"]>50})
PY
