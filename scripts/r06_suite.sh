#!/bin/bash
# the whole GPU suite (also with the dead rows' buffers poisoned) + the default bench line
TAG=${1:-r06s}
mkdir -p gpurun_out/$TAG
timeout 2400 python -m pytest tests -q -m gpu > gpurun_out/$TAG/suite.log 2>&1
echo "rc=$?" >> gpurun_out/$TAG/suite.log
DMP_POISON_DEAD_ROWS=1 timeout 2400 python -m pytest tests -q -m gpu > gpurun_out/$TAG/suite_poisoned.log 2>&1
echo "rc=$?" >> gpurun_out/$TAG/suite_poisoned.log
tail -6 gpurun_out/$TAG/suite.log | cut -c1-400
tail -6 gpurun_out/$TAG/suite_poisoned.log | cut -c1-400
timeout 900 python bench.py > gpurun_out/$TAG/bench_line.json 2> gpurun_out/$TAG/bench_err.txt
python - <<PY
import json
d = json.loads([l for l in open("gpurun_out/$TAG/bench_line.json") if l.startswith("{")][-1])
print("value", d["value"], "ms", d["ms_per_step"], "eager", d["eager_ms_per_step"], "all_out", d["all_outputs_ms_per_step"])
print("roofline", d["roofline"]["frac"], d["roofline"]["avg_us"], d["roofline"].get("same_size_copy"))
print("roofline_bwd", d["roofline_bwd"]["frac"], d["roofline_bwd"]["avg_us"], d["roofline_bwd"].get("same_size_copy"))
gd = d["gate_dense"]; print("gate_dense", gd["ms_per_step"], gd["roofline"]["frac"], gd["roofline"]["avg_us"], gd["roofline_bwd"]["frac"], gd["roofline_bwd"]["avg_us"])
print("gate_compact", d["gate_compact"]["ms_per_step"], "cpu", d["cpu_baseline"]["value"])
PY
