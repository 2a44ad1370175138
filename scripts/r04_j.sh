#!/bin/bash
mkdir -p gpurun_out/r04i
DMP_POISON_DEAD_ROWS=1 timeout 1500 python -m pytest tests/test_gpu_compact.py tests/test_gpu_dmplayer.py tests/test_gpu_layer0.py tests/test_gpu_bench_composite.py tests/test_gpu_kernels.py tests/test_gpu_fullmodel.py tests/test_gpu_bf16x6.py -x -q -m gpu > gpurun_out/r04i/tests.log 2>&1
echo "rc=$?" >> gpurun_out/r04i/tests.log
tail -5 gpurun_out/r04i/tests.log | cut -c1-300
timeout 500 python3 bench.py --no-cpu-baseline --no-all-outputs --extended-steps 0 > gpurun_out/r04i/bench.json 2> gpurun_out/r04i/bench.err
DMP_PLAIN_ATB=0 timeout 500 python3 bench.py --no-cpu-baseline --no-all-outputs --no-gate-compact --extended-steps 0 > gpurun_out/r04i/bench_off.json 2> gpurun_out/r04i/bench.err
python3 - <<'PY'
import json
for n in ("bench","bench_off"):
    p=json.loads([l for l in open("gpurun_out/r04i/%s.json"%n) if l.startswith("{")][-1])
    print(n, p["value"], p["ms_per_step"], p["step_ms_median"], p.get("gate_compact") and p["gate_compact"]["ms_per_step"])
    for k,v in sorted(p["kernels"].items()):
        if any(x in k for x in ("atb_rows","smallk_atb[K=1,")): print("   ", k, v["avg_us"], v["launches"])
PY
