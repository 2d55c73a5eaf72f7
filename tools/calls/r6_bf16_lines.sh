R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; OUT=$R/gpurun_out/prof_r06; mkdir -p $OUT
timeout 600 python $R/bench.py --steps 40 --warmup 8 --cpu-images 0 --from-rgb-steps 0 --latency-reps 0 --stress-steps 0 --lbs-unfused-reps 0 --mesh-arith bf16x3 > $OUT/bench_bf16x3.log 2>&1
grep -h '^{' $OUT/bench_bf16x3.log | tail -1 > $OUT/bench_bf16x3.json
timeout 300 python $R/bench.py --live-traffic off --batch 16 --num-samples 1000 --steps 10 --warmup 3 --cpu-images 0 --from-rgb-steps 0 --latency-reps 0 --stress-steps 0 --lbs-unfused-reps 0 --mesh-arith bf16x3 > $OUT/bench_n1000_bf16x3.log 2>&1
grep -h '^{' $OUT/bench_n1000_bf16x3.log | tail -1 > $OUT/bench_n1000_bf16x3.json
python -c "
import json
for n in ('bench_bf16x3','bench_n1000_bf16x3'):
    d=json.load(open('$OUT/%s.json'%n)); r=d['roofline']; print(n, round(d['value']), round(d['ms_per_step'],3), r['kernel'], round(r['avg_launch_ms'],4), 'traffic', r['traffic'], r['traffic_imported'], d['config']['mesh_arith'], d['config']['step_pipelining'][-90:])"
