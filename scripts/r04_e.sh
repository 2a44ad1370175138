#!/bin/bash
mkdir -p gpurun_out/r04e
timeout 600 python scripts/bf16x6_probe.py > gpurun_out/r04e/bf16x6.json 2> gpurun_out/r04e/bf16x6.err
cat gpurun_out/r04e/bf16x6.json; tail -5 gpurun_out/r04e/bf16x6.err
