#!/bin/bash
# the whole GPU suite (also with the dead rows' buffers poisoned) + the default bench line
TAG=${1:-r05s}
mkdir -p gpurun_out/$TAG
timeout 1500 python -m pytest tests -q -m gpu > gpurun_out/$TAG/suite.log 2>&1
echo "rc=$?" >> gpurun_out/$TAG/suite.log
DMP_POISON_DEAD_ROWS=1 timeout 1500 python -m pytest tests -q -m gpu > gpurun_out/$TAG/suite_poisoned.log 2>&1
echo "rc=$?" >> gpurun_out/$TAG/suite_poisoned.log
tail -4 gpurun_out/$TAG/suite.log | cut -c1-300
tail -4 gpurun_out/$TAG/suite_poisoned.log | cut -c1-300
