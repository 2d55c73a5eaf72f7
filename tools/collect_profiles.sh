#!/bin/bash
# Run ON THE GPU BOX (through gpurun): everything the committed profiles/ summaries are made of.
#   1. rocprofv3 kernel stats of the default bench command (12 steps profiled)
#   2. separate --pmc passes (FETCH_SIZE, WRITE_SIZE do not fit one pass) for the mesh kernel inside the bench
#   3. stand-alone PMC of the fused mesh kernel (tools/mesh_pmc.sh) and per-layer encoder table (tools/encoder_layers.py / encoder_pmc.sh)
#   4. the BASELINE configs[4] stress configuration (B = 16, N = 1000) bench line and its kernel stats
# Raw output under gpurun_out/prof_$TAG/; tools/summarize_profiles.py $TAG reduces it to profiles/.
# usage: tools/collect_profiles.sh r02
TAG=${1:-r04}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
MESH="mesh_fused_kernel"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o bench -- python $R/bench.py --live-traffic off --steps 10 --warmup 2 --cpu-images 0 --from-rgb-steps 0 --latency-reps 0 --stress-steps 0 > $OUT/bench_stats.log 2>&1
echo "kernel stats exit $?"
grep -h '^{' $OUT/bench_stats.log | tail -1 > $OUT/bench_under_rocprof.json
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $c --kernel-trace --kernel-include-regex "$MESH" --output-format csv -d $OUT -o mesh_$c -- python $R/bench.py --live-traffic off --steps 3 --warmup 1 --cpu-images 0 --from-rgb-steps 0 --latency-reps 0 --stress-steps 0 --lbs-unfused-reps 0 > $OUT/pmc_$c.log 2>&1
  echo "pmc $c exit $?"
done
for k in uncertainty_joints_kernel mf_sample_kernel; do
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 300 rocprofv3 --pmc $c --kernel-trace --kernel-include-regex "$k" --output-format csv -d $OUT -o ${k}_$c -- python $R/bench.py --live-traffic off --steps 3 --warmup 1 --cpu-images 0 --from-rgb-steps 0 --latency-reps 0 --stress-steps 0 --lbs-unfused-reps 0 --no-pipeline > $OUT/pmc_${k}_$c.log 2>&1
    echo "pmc $k $c exit $?"
  done
done
# the stress configuration (BASELINE configs[4]): bench line + kernel stats
timeout 300 python $R/bench.py --live-traffic off --batch 16 --num-samples 1000 --steps 10 --warmup 3 --cpu-images 0 --latency-reps 0 --stress-steps 0 > $OUT/bench_n1000.log 2>&1
echo "configs[4] bench exit $?"
grep -h '^{' $OUT/bench_n1000.log | tail -1 > $OUT/bench_n1000.json
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o n1000 -- python $R/bench.py --live-traffic off --batch 16 --num-samples 1000 --steps 6 --warmup 2 --cpu-images 0 --from-rgb-steps 0 --latency-reps 0 --stress-steps 0 > $OUT/n1000_stats.log 2>&1
echo "configs[4] kernel stats exit $?"
# the opt-in bf16x3 arithmetic of the mesh kernel as the headline's kernel (B = 64, N = 100, live PMC traffic of mesh_split_kernel) and at configs[4]
timeout 600 python $R/bench.py --steps 40 --warmup 8 --cpu-images 0 --from-rgb-steps 0 --latency-reps 0 --stress-steps 0 --lbs-unfused-reps 0 --mesh-arith bf16x3 > $OUT/bench_bf16x3.log 2>&1
grep -h '^{' $OUT/bench_bf16x3.log | tail -1 > $OUT/bench_bf16x3.json
timeout 300 python $R/bench.py --live-traffic off --batch 16 --num-samples 1000 --steps 10 --warmup 3 --cpu-images 0 --from-rgb-steps 0 --latency-reps 0 --stress-steps 0 --lbs-unfused-reps 0 --mesh-arith bf16x3 > $OUT/bench_n1000_bf16x3.log 2>&1
grep -h '^{' $OUT/bench_n1000_bf16x3.log | tail -1 > $OUT/bench_n1000_bf16x3.json
# the unfused definition (blend GEMM + LBS as two kernels) for comparison, and the non-pipelined loop
timeout 300 python $R/bench.py --live-traffic off --steps 20 --warmup 5 --cpu-images 0 --from-rgb-steps 0 --latency-reps 0 --stress-steps 0 --unfused-mesh > $OUT/bench_unfused.log 2>&1
grep -h '^{' $OUT/bench_unfused.log | tail -1 > $OUT/bench_unfused.json
timeout 300 python $R/bench.py --live-traffic off --steps 20 --warmup 5 --cpu-images 0 --from-rgb-steps 0 --latency-reps 0 --stress-steps 0 --no-pipeline > $OUT/bench_nopipe.log 2>&1
grep -h '^{' $OUT/bench_nopipe.log | tail -1 > $OUT/bench_nopipe.json
# the default command, un-profiled, with the CPU baseline: the line the driver will measure
timeout 600 python $R/bench.py > $OUT/bench_default.log 2>&1
echo "default bench exit $?"
grep -h '^{' $OUT/bench_default.log | tail -1 > $OUT/bench_line.json
# the driver's flags (--steps 20 --warmup 5): fill and drain of the pipeline weigh 1/20 instead of 1/40
timeout 600 python $R/bench.py --steps 20 --warmup 5 > $OUT/bench_driver_flags.log 2>&1
grep -h '^{' $OUT/bench_driver_flags.log | tail -1 > $OUT/bench_driver_flags.json
# batch-1 latency (both encoder modes) with the kernel timeline of one call in latency mode
timeout 120 python $R/tools/latency_b1.py 40 > $OUT/latency_b1.txt 2>&1
timeout 120 python $R/tools/latency_b1.py 40 --latency >> $OUT/latency_b1.txt 2>&1
timeout 200 rocprofv3 --kernel-trace --output-format csv -d $OUT -o lat -- python $R/tools/latency_b1.py 12 --latency > $OUT/lat_trace.log 2>&1
python $R/tools/latency_b1.py analyse $(find $OUT -name "lat_kernel_trace.csv" | head -1) > $OUT/latency_b1_timeline.txt 2>&1
# the widened rows and the predict loop
timeout 200 python $R/tools/next_rows_time.py > $OUT/next_rows.txt 2>&1
for a in "4096 64 50" "2048 16 50" "1024 1 50" "1024 1 50 --latency" "1024 2 50 --latency" "2048 64 50 --pageable"; do timeout 200 python $R/tools/predict_time.py $a 2>&1 | tail -1 >> $OUT/predict_time.txt; done
# SQ / LDS counters of the fused mesh kernel
bash $R/tools/mesh_pmc_lds.sh > $OUT/mesh_pmc_lds.txt 2>&1
