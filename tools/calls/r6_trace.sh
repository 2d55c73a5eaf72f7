#!/bin/bash
# kernel trace of the default bench loop: every kernel of one steady-state step in order (tools/inloop_vs_alone.py) and by name (tools/step_launches.py)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
OUT=$R/gpurun_out/r6_trace; rm -rf $OUT; mkdir -p $OUT
Q="--cpu-images 0 --from-rgb-steps 0 --stress-steps 0 --latency-reps 0 --lbs-unfused-reps 0 --live-traffic off"
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o bench -- python $R/bench.py --steps 12 --warmup 4 $Q "$@" > $OUT/trace_log.txt 2>&1
cd $R
T=$(find $OUT -name "*kernel_trace.csv" | head -1)
python tools/step_launches.py trace $T > $OUT/step_trace.txt 2>&1; cat $OUT/step_trace.txt
python tools/inloop_vs_alone.py $T > $OUT/inloop.txt 2>&1; cat $OUT/inloop.txt
python tools/step_launches.py aten > $OUT/step_aten.txt 2>&1; grep -c LAUNCH $OUT/step_aten.txt; head -3 $OUT/step_aten.txt
find $OUT -name "*kernel_trace.csv" -size +20M -delete
