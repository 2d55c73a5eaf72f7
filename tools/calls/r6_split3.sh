cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6_split
for rep in 1 2 3; do
python bench.py --steps 20 --warmup 5 --cpu-images 0 --from-rgb-steps 0 --latency-reps 0 --stress-steps 0 --lbs-unfused-reps 0 --live-traffic off --split-steps 0 --mesh-arith bf16x3 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('bf16x3', round(d['value']), 'images/s', round(d['ms_per_step'],3), 'ms; mesh', round(d['roofline']['avg_launch_ms'],4), 'enc', round(d['secondary']['encoder']['avg_ms'],3))"
python bench.py --steps 20 --warmup 5 --cpu-images 0 --from-rgb-steps 0 --latency-reps 0 --stress-steps 0 --lbs-unfused-reps 0 --live-traffic off --split-steps 0 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('f32   ', round(d['value']), 'images/s', round(d['ms_per_step'],3), 'ms; mesh', round(d['roofline']['avg_launch_ms'],4), 'enc', round(d['secondary']['encoder']['avg_ms'],3))"
done
