#!/bin/bash
mkdir -p gpurun_out/r04i
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_dmplayer.py tests/test_gpu_compact.py -x -q -m gpu > gpurun_out/r04i/tests.log 2>&1
echo "rc=$?" >> gpurun_out/r04i/tests.log
tail -4 gpurun_out/r04i/tests.log | cut -c1-300
for v in 0 1; do
DMP_ATB_ROWS_X6=$v timeout 500 python3 bench.py --no-cpu-baseline --no-all-outputs --no-gate-compact --extended-steps 0 > gpurun_out/r04i/bench_$v.json 2> gpurun_out/r04i/bench.err
done
DMP_ROW_MASKS=0 timeout 500 python3 bench.py --no-cpu-baseline --no-all-outputs --no-gate-compact --extended-steps 0 > gpurun_out/r04i/bench_nomask.json 2> gpurun_out/r04i/bench.err
python3 - <<'PY'
import json
for n in ("bench_0","bench_1","bench_nomask"):
    p=json.loads([l for l in open("gpurun_out/r04i/%s.json"%n) if l.startswith("{")][-1])
    print(n, p["value"], p["ms_per_step"], p["step_ms_median"])
    for k,v in sorted(p["kernels"].items()):
        if any(x in k for x in ("out_fwd_mfma[H=128,R=548864","bwd_h1_mfma[H=128,E=548864","atb_rows[M=128,N=128,R=548864")): print("   ", k, v["avg_us"])
PY
