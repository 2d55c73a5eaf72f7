cd $GRAFT_REPO_ROOT
Q="--steps 20 --warmup 5 --cpu-images 0 --from-rgb-steps 0 --stress-steps 0 --latency-reps 0 --lbs-unfused-reps 0 --live-traffic off"
for rep in 1 2; do for v in "" "--head-high-priority"; do for cfg in "--batch 16 --num-samples 1000" "--batch 16 --num-samples 100" "--batch 8 --num-samples 50"; do python bench.py $Q $cfg $v 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('[%-32s %-24s] %6d images/s  %.3f ms/step' % ('$cfg', '$v', d['value'], d['ms_per_step']))"; done; done; done
