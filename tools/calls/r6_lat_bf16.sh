cd $GRAFT_REPO_ROOT
for a in f32 bf16x3 f32 bf16x3; do
python bench.py --steps 6 --warmup 3 --cpu-images 0 --live-traffic off --from-rgb-steps 0 --stress-steps 0 --lbs-unfused-reps 0 --split-steps 0 --mesh-arith $a 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); l=d['secondary']['latency_b1']; print('$a: latency_b1 median', round(l['median_ms'],4), 'graph', round(l['graph_median_ms'],4), 'throughput-mode', round(l['throughput_mode_median_ms'],4))"
done
python -m pytest tests/test_gpu_smpl.py tests/test_gpu_e2e.py -x -q -m gpu -k "split or bf16x3" 2>&1 | tail -2
