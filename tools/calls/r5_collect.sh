#!/bin/bash
# Round 5, run ON THE GPU BOX through gpurun: everything profiles/r05_* is made of.
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
bash tools/collect_profiles.sh r05 2>&1 | tail -12
OUT=$R/gpurun_out/prof_r05
# per-layer encoder table (alone / in the list) + its PMC pass
python tools/encoder_layers.py time > $OUT/encoder_layers_time.log 2>&1
bash tools/encoder_pmc.sh > $OUT/encoder_pmc.log 2>&1
# round-5 A/Bs quoted in DESIGN.md / profiles/r05_experiments.txt
{
echo "==== tests/dev/pair_time.py (hps_smpl_pose_prep / hps_smpl_joints against their first generations, alone, vertices from HBM) ===="
python tests/dev/pair_time.py 2>&1 | grep "M ="
echo "==== tests/dev/wino_ks_time.py 64 / 16 (K slices of the Winograd 8 x 8 geometry) ===="
python tests/dev/wino_ks_time.py 64 2>&1 | grep "K slices"
python tests/dev/wino_ks_time.py 16 2>&1 | grep "K slices"
echo "==== tools/latency_b1.py 40 --latency [--per-level] (batch-1 infer(): eight level launches against the single-launch experiment) ===="
for i in 1 2; do python tools/latency_b1.py 40 --latency --per-level 2>&1 | tail -1; python tools/latency_b1.py 40 --latency 2>&1 | tail -1; done
echo "==== bench.py --from-rgb-variant default | nchw (front end writes the stem's phase frames directly against the NCHW tensor + phase split) ===="
for v in default nchw default nchw; do python bench.py --steps 20 --warmup 5 --cpu-images 0 --stress-steps 0 --latency-reps 0 --lbs-unfused-reps 0 --from-rgb-variant $v 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['secondary']['from_rgb']; print('$v: headline', round(d['value']), 'from_rgb legs', [round(x) for x in r['legs_images_per_s']], 'encoder ms', round(r['encoder_avg_ms'],3), 'mesh kernel ms', round(r['mesh_kernel_avg_ms'],3))"; done
echo "==== tests/dev/wino8_check.py (eight-wave Winograd kernel: four-wave / lane-per-tile forms, ablations, s_setprio) ===="
python tests/dev/wino8_check.py 2>&1 | grep -v "identical=True" | grep -v amdgpu.ids
echo "==== tests/dev/wino_phases.py (shader-clock stamps between two items) ===="
python tests/dev/wino_phases.py 2>&1 | grep -v amdgpu.ids
echo "==== tests/dev/wino_half_check.py (two four-wave workgroups per CU on half items) ===="
python tests/dev/wino_half_check.py 2>&1 | grep -v amdgpu.ids | grep -v "identical=True"
python tests/dev/wino_half_check.py 2>&1 | grep -c "identical=True" | sed 's/^/   configurations bit-identical to the product: /'
echo "==== tests/dev/marker_cost.py (an event record between two kernels of a stream) ===="
python tests/dev/marker_cost.py 2>&1 | grep -v amdgpu.ids
echo "==== bench.py [--no-inline-mesh] [--event-every 0], 40 steps ===="
for i in 1 2 3; do for v in "" "--no-inline-mesh" "--event-every 0"; do python bench.py --steps 40 --warmup 10 --cpu-images 0 --from-rgb-steps 0 --stress-steps 0 --latency-reps 0 --lbs-unfused-reps 0 $v 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('[$v]', round(d['value']), 'images/s', round(d['ms_per_step'],3), 'ms')"; done; done
} > $OUT/ab.txt 2>&1
bash tools/calls/next_rows_stats.sh > $OUT/next_rows_stats.log 2>&1
cat $OUT/ab.txt
