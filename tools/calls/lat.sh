R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/lat; rm -rf $OUT; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
timeout 200 python -m pytest $R/tests/test_gpu_net.py -m gpu -x -q -k "latency_mode or winograd_and_direct" 2>&1 | tail -3
timeout 120 python $R/tools/latency_b1.py 40
timeout 120 python $R/tools/latency_b1.py 40 --latency
timeout 200 rocprofv3 --kernel-trace --output-format csv -d $OUT -o lat -- python $R/tools/latency_b1.py 12 --latency > $OUT/log.txt 2>&1
echo "exit $?"
f=$(find $OUT -name "*kernel_trace.csv" | head -1)
python $R/tools/latency_b1.py analyse $f > $OUT/timeline.txt; tail -3 $OUT/timeline.txt
