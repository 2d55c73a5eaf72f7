cd $GRAFT_REPO_ROOT
Q="--steps 40 --warmup 10 --cpu-images 0 --from-rgb-steps 0 --stress-steps 0 --latency-reps 0 --lbs-unfused-reps 0 --live-traffic off --split-steps 0 --mesh-arith bf16x3"
for rep in 1 2 3; do for m in 0 13; do python tests/dev/ab_bench.py hps_dev_mesh_split_ablate $m $Q 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('ablate $m: %6d images/s  %.3f ms/step  mesh %.4f' % (d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms']))"; done; done
