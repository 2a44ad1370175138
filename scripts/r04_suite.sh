#!/bin/bash
# the whole GPU suite (also with the dead rows' buffers poisoned) + the default bench line
mkdir -p gpurun_out/r04s
timeout 2400 python -m pytest tests -q -m gpu > gpurun_out/r04s/suite.log 2>&1
echo "rc=$?" >> gpurun_out/r04s/suite.log
DMP_POISON_DEAD_ROWS=1 timeout 2400 python -m pytest tests -q -m gpu > gpurun_out/r04s/suite_poisoned.log 2>&1
echo "rc=$?" >> gpurun_out/r04s/suite_poisoned.log
tail -3 gpurun_out/r04s/suite.log | cut -c1-200
tail -3 gpurun_out/r04s/suite_poisoned.log | cut -c1-200
