mkdir -p gpurun_out/r05h
DMP_POISON_DEAD_ROWS=1 timeout 900 python -m pytest tests/test_gpu_pins_r2.py tests/test_gpu_harness.py tests/test_gpu_layer0.py -q -x > gpurun_out/r05h/pins.log 2>&1; echo "pins poisoned rc=$?"; tail -3 gpurun_out/r05h/pins.log | cut -c1-300
timeout 900 python -m pytest tests/test_gpu_ragged_replay.py -q -x > gpurun_out/r05h/ragged.log 2>&1; echo "ragged rc=$?"; tail -15 gpurun_out/r05h/ragged.log | cut -c1-300
for i in 1 2 3; do timeout 600 python -m pytest tests/test_gpu_bench_line.py -q -x -k "one_rank_rccl" > gpurun_out/r05h/rccl_$i.log 2>&1; echo "rccl $i rc=$?"; done
timeout 600 python scripts/kbench_train_ragged.py > gpurun_out/r05h/ragged_bench.txt 2>&1; tail -9 gpurun_out/r05h/ragged_bench.txt
timeout 600 python scripts/kbench_atb.py > gpurun_out/r05h/atb.json 2>&1; cat gpurun_out/r05h/atb.json | tail -25
