#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/trace_r03; rm -rf $OUT; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT -o bench -- python $R/bench.py --steps 10 --warmup 3 --cpu-images 0 --lbs-unfused-reps 0 $@ > $OUT/log.txt 2>&1
echo "exit $?"; grep -h '^{' $OUT/log.txt | tail -1 | cut -c1-160
python3 $R/tools/inloop_vs_alone.py $(ls $OUT/*kernel_trace.csv | head -1)
