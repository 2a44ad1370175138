cd /tmp && export TMPDIR=/tmp && R=$GRAFT_REPO_ROOT && O=$R/gpurun_out/r02p && mkdir -p $O
python3 $R/bench.py > $O/bench_line.json 2> $O/bench_line.err
python3 $R/bench.py --act relu --emb Orthogonal --no-cpu-baseline > $O/bench_relu_line.json 2> /dev/null
python3 $R/bench.py --workload 4 --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_c4_line.json 2> /dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o b -- python3 $R/bench.py --no-cpu-baseline > $O/prof_bench.json 2> $O/prof_err.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof4 -o c4 -- python3 $R/bench.py --workload 4 --steps 6 --warmup 2 --no-cpu-baseline > $O/prof_c4.json 2>> $O/prof_err.txt
python3 $R/bench.py --hid 64 --no-cpu-baseline > $O/bench_h64_line.json 2> /dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof64 -o h64 -- python3 $R/bench.py --hid 64 --no-cpu-baseline > $O/prof_h64.json 2>> $O/prof_err.txt
for k in in inc; do
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_f_$k -o f -- python3 $R/scripts/prof_seg_sum2.py 10 $k > /dev/null 2>> $O/prof_err.txt
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_w_$k -o w -- python3 $R/scripts/prof_seg_sum2.py 10 $k > /dev/null 2>> $O/prof_err.txt
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_$k -o k -- python3 $R/scripts/prof_seg_sum2.py 20 $k > /dev/null 2>> $O/prof_err.txt
done
cd $R && python3 scripts/host_time.py > $O/host_time.txt 2>&1; HID=64 python3 scripts/host_time.py > $O/host_time_h64.txt 2>&1; python3 scripts/kbench_rgnn.py > $O/rgnn.txt 2>&1
python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 10 --warmup 3 --backend gloo --single-device > $O/bench_gloo2_line.json 2> $O/bench_gloo2.err
rm -f $O/prof/*trace* $O/prof4/*trace* $O/prof64/*trace*; du -sh $O; ls $O $O/prof $O/pmc_f_in | head -40
