R=$GRAFT_REPO_ROOT; cd $R
timeout 600 python -m pytest tests/test_gpu_net.py tests/test_gpu_e2e.py -m gpu -x -q 2>&1 | tail -3
timeout 120 python tools/latency_b1.py 40 --latency
timeout 120 python bench.py --steps 20 --warmup 5 --cpu-images 0 --lbs-unfused-reps 0 --from-rgb-steps 0 --latency-reps 0 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']), d['ms_per_step'], d['metric_checksums'])"
