#!/bin/bash
mkdir -p gpurun_out/r04i
timeout 1500 python -m pytest tests/test_gpu_compact.py tests/test_gpu_kernels.py tests/test_gpu_dmplayer.py tests/test_gpu_bench_composite.py tests/test_gpu_bf16x6.py tests/test_gpu_shapes.py tests/test_gpu_layer0.py tests/test_gpu_fullmodel.py tests/test_gpu_lazy_rows.py -x -q -m gpu > gpurun_out/r04i/tests.log 2>&1
echo "rc=$?" >> gpurun_out/r04i/tests.log
tail -12 gpurun_out/r04i/tests.log | cut -c1-300
timeout 500 python3 bench.py --no-cpu-baseline --no-all-outputs > gpurun_out/r04i/bench.json 2> gpurun_out/r04i/bench.err
python3 - <<'PY'
import json
for n in ("bench",):
    p=json.loads([l for l in open("gpurun_out/r04i/%s.json"%n) if l.startswith("{")][-1])
    print(n, p["value"], p["ms_per_step"], p["step_ms_median"], "gate:", p["gate_compact"]["ms_per_step"])
    for k,v in sorted(p["kernels"].items()):
        if any(x in k for x in ("out_fwd_mfma[","bwd_h1_mfma[","atb_rows[M=128,N=128,R=548864")): print("   ", k, v["avg_us"])
PY
