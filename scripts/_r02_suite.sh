R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02s; mkdir -p $O
python -m pytest tests -x -q -m gpu > $O/pytest.txt 2>&1; tail -5 $O/pytest.txt
python bench.py --hid 64 --no-cpu-baseline 2>/dev/null > $O/bench_h64.json; cut -c1-330 $O/bench_h64.json
python bench.py --no-cpu-baseline 2>/dev/null > $O/bench_h128.json; cut -c1-330 $O/bench_h128.json
HID=64 python3 scripts/host_time.py 2>&1 | tail -4
