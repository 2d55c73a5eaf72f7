cd $GRAFT_REPO_ROOT
show() { python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$1:', round(d['value']), 'images/s', round(d['ms_per_step'],3), 'ms; mesh', round(d['roofline']['avg_launch_ms'],3), 'enc', round(d['secondary']['encoder']['avg_ms'],3))"; }
Q="--batch 16 --num-samples 1000 --steps 12 --warmup 4 --cpu-images 0 --live-traffic off --from-rgb-steps 0 --latency-reps 0 --lbs-unfused-reps 0 --stress-steps 0 --split-steps 0"
for rep in 1 2; do
for a in f32 bf16x3; do for k in 0 8 12 16 20; do
python bench.py $Q --mesh-arith $a --encoder-cus $k 2>/dev/null | show "$a encoder_cus=$k"
done; done; done
