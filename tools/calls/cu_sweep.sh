cd $GRAFT_REPO_ROOT
for rep in 1 2; do for c in 0 4 6 8 10; do python bench.py --batch 16 --num-samples 1000 --steps 12 --warmup 3 --cpu-images 0 --lbs-unfused-reps 0 --mesh-overlap on --encoder-cus $c 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('B16 N1000 encoder-cus $c  %6d images/s  %.3f ms/step  encoder %.3f ms  mesh kernel %.3f ms' % (d['value'], d['ms_per_step'], d['secondary']['encoder']['avg_ms'], d['roofline']['avg_launch_ms']))"; done; done
for c in 0 8 16 24; do python bench.py --steps 20 --warmup 5 --cpu-images 0 --lbs-unfused-reps 0 --mesh-overlap on --encoder-cus $c 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('B64 N100 overlap on encoder-cus $c  %6d images/s  %.3f ms/step  encoder %.3f ms  mesh kernel %.3f ms' % (d['value'], d['ms_per_step'], d['secondary']['encoder']['avg_ms'], d['roofline']['avg_launch_ms']))"; done
