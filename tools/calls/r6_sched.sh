#!/bin/bash
# Round 6, call 1 (run ON THE GPU BOX through gpurun): the schedule A/B of VERDICT r5 item 1 -- stem route (NCHW-fed / phase frames) x
# side kernels (behind the mesh kernel on the encoder's stream / on the caller's stream) x the bench's own timing events (on / off),
# interleaved on one box -- and the launches of one steady-state step, named.
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
OUT=$R/gpurun_out/r6_sched; rm -rf $OUT; mkdir -p $OUT
Q="--cpu-images 0 --from-rgb-steps 0 --stress-steps 0 --latency-reps 0 --lbs-unfused-reps 0 --live-traffic off"
one() { python bench.py --steps 40 --warmup 10 $Q "$@" 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); e=d['secondary'].get('encoder',{}); print('%-62s %6d images/s  %.3f ms/step  encoder %s  mesh %s' % ('[$*]', d['value'], d['ms_per_step'], ('%.3f' % e['avg_ms']) if e else '  -  ', ('%.3f' % d['roofline']['avg_launch_ms']) if d['roofline']['launches'] else '  -  '))"; }
{
for rep in 1 2 3; do
  for ev in 1 0; do
    one --stem-route frames --side-on-caller-stream --event-every $ev
    one --stem-route frames --event-every $ev
    one --stem-route nchw --event-every $ev
    one --stem-route nchw --side-on-caller-stream --event-every $ev
  done
done
} > $OUT/ab.txt 2>&1
cat $OUT/ab.txt
python tools/step_launches.py aten > $OUT/step_aten.txt 2>&1; tail -40 $OUT/step_aten.txt
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT -o bench -- python $R/bench.py --steps 10 --warmup 3 $Q --stem-route nchw > $OUT/trace_log.txt 2>&1
cd $R
python tools/step_launches.py trace $(find $OUT -name "*kernel_trace.csv" | head -1) > $OUT/step_trace.txt 2>&1; cat $OUT/step_trace.txt
python tools/inloop_vs_alone.py $(find $OUT -name "*kernel_trace.csv" | head -1) > $OUT/inloop.txt 2>&1; head -70 $OUT/inloop.txt
find $OUT -name "*kernel_trace.csv" -size +20M -delete
