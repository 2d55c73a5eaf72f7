R=$GRAFT_REPO_ROOT; cd $R
timeout 300 python -m pytest tests/test_gpu_net.py -m gpu -x -q -k "latency" 2>&1 | tail -3
timeout 120 python tools/latency_b1.py 40
timeout 120 python tools/latency_b1.py 40 --latency
OUT=$R/gpurun_out/lat; rm -rf $OUT; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --output-format csv -d $OUT -o lat -- python $R/tools/latency_b1.py 12 --latency > $OUT/log.txt 2>&1
python $R/tools/latency_b1.py analyse $(find $OUT -name "*kernel_trace.csv" | head -1) | grep -E "joint_level|linear|kernels "
