#!/bin/bash
# the whole GPU suite + the default bench line
mkdir -p gpurun_out/r04s
timeout 2400 python -m pytest tests -q -m gpu > gpurun_out/r04s/suite.log 2>&1
echo "rc=$?" >> gpurun_out/r04s/suite.log
timeout 600 python bench.py > gpurun_out/r04s/bench.json 2> gpurun_out/r04s/bench.err
tail -15 gpurun_out/r04s/suite.log
