#!/bin/bash
mkdir -p gpurun_out/r04i
timeout 1500 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_dmplayer.py tests/test_gpu_compact.py tests/test_gpu_layer0.py tests/test_gpu_lazy_rows.py tests/test_gpu_bench_composite.py tests/test_gpu_fullmodel.py -x -q -m gpu > gpurun_out/r04i/tests.log 2>&1
echo "rc=$?" >> gpurun_out/r04i/tests.log
tail -4 gpurun_out/r04i/tests.log | cut -c1-300
timeout 500 python3 bench.py --no-cpu-baseline --no-all-outputs --extended-steps 0 > gpurun_out/r04i/bench.json 2> gpurun_out/r04i/bench.err

python3 - <<'PY'
import json
for n in ("bench",):
    p=json.loads([l for l in open("gpurun_out/r04i/%s.json"%n) if l.startswith("{")][-1])
    print(n, p["value"], p["ms_per_step"], p["step_ms_median"], p.get("gate_compact") and p["gate_compact"]["ms_per_step"])
    for k,v in sorted(p["kernels"].items()):
        if any(x in k for x in ("smallk_atb","l0_bwd_w[K=10,E=524288")): print("   ", k, v["avg_us"])
PY
