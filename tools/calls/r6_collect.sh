#!/bin/bash
# Round 6, run ON THE GPU BOX through gpurun: everything profiles/r06_* is made of.
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
bash tools/collect_profiles.sh r06 2>&1 | tail -14
OUT=$R/gpurun_out/prof_r06
# per-layer encoder table (alone / in the list) + its PMC pass
python tools/encoder_layers.py time > $OUT/encoder_layers_time.log 2>&1
bash tools/encoder_pmc.sh > $OUT/encoder_pmc.log 2>&1
# one steady-state step: every launch by name (kernel trace) and every ATen operator with its call site
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT -o steptrace -- python $R/bench.py --steps 12 --warmup 4 --cpu-images 0 --from-rgb-steps 0 --stress-steps 0 --latency-reps 0 --lbs-unfused-reps 0 > $OUT/steptrace.log 2>&1
cd $R
T=$(find $OUT -name "steptrace_kernel_trace.csv" | head -1)
{ python tools/step_launches.py trace $T; echo; python tools/step_launches.py aten 2>/dev/null | grep -v "no launch"; } > $OUT/step_launches.txt 2>&1
python tools/inloop_vs_alone.py $T > $OUT/step_timeline.txt 2>&1
rm -f $T
# round-6 A/Bs, interleaved on this box: the default against each round-5 setting it replaced, and against its own run without the bench's timing events
Q="--steps 40 --warmup 10 --cpu-images 0 --from-rgb-steps 0 --stress-steps 0 --latency-reps 0 --lbs-unfused-reps 0 --live-traffic off"
one() { python bench.py $Q "$@" 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); e=d['secondary'].get('encoder',{}); r=d['roofline']; print('%-64s %6d images/s  %.3f ms/step  encoder %s  mesh %s' % ('[$*]', d['value'], d['ms_per_step'], ('%.3f' % e['avg_ms']) if e else '  -  ', ('%.4f' % r['avg_launch_ms']) if r['launches'] else '  -  '))"; }
{
echo "==== bench.py $Q <flags>: three interleaved rounds ===="
for rep in 1 2 3; do
  one
  one --event-every 0
  one --stem-route frames --side-on-caller-stream
  one --stem-route nchw --side-on-caller-stream --event-every 0
  one --separate-downsample
  one --per-mesh-shape-blend
  one --separate-joints
  one --stem-route frames --side-on-caller-stream --separate-downsample --per-mesh-shape-blend --separate-joints
done
echo "==== tests/dev/ab_bench.py hps_dev_unc_mode {0 | 7}: the uncertainty pass's block order (7 = round 5's: panels fastest, ascending) ===="
for rep in 1 2 3; do for m in 0 7; do python tests/dev/ab_bench.py hps_dev_unc_mode $m $Q --separate-joints 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('unc mode $m: %6d images/s  %.3f ms/step' % (d['value'], d['ms_per_step']))"; done; done
} > $OUT/ab.txt 2>&1
bash tools/calls/next_rows_stats.sh > $OUT/next_rows_stats.log 2>&1
cat $OUT/ab.txt
# the opt-in bf16x3 arithmetic of the mesh kernel: interleaved A/B of the headline, the kernel alone with its ablations, counters, the bf16 MFMA ceiling
{
echo "==== bench.py $Q --split-steps 0 [--mesh-arith bf16x3]: three interleaved pairs ===="
for rep in 1 2 3; do one --split-steps 0; one --split-steps 0 --mesh-arith bf16x3; done
echo "==== tests/dev/mesh_split_time.py --ablations (6 528 meshes, kernel alone) ===="
python tests/dev/mesh_split_time.py --ablations 2>&1 | grep -v amdgpu.ids
echo "==== tools/bin/mfma_bf16_peak ===="
tools/bin/mfma_bf16_peak
echo "==== tools/mesh_split_pmc.sh ===="
bash tools/mesh_split_pmc.sh 2>&1
} > $OUT/mesh_bf16x3.txt 2>&1
cat $OUT/mesh_bf16x3.txt | tail -60
