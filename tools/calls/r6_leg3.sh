cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_e2e.py -x -q -m gpu -k "bench_line or partition or stress or bf16x3" 2>&1 | tail -4
show() { python -c "
import json,sys; d=json.loads(sys.stdin.read()); s=d['secondary']['stress_n1000']; print('$1: stress f32', round(s['images_per_s']), round(s['ms_per_step'],3), 'mesh', round(s['mesh_kernel']['median_ms'],3), '| bf16x3', s['mesh_bf16x3'])"; }
for rep in 1 2; do
python bench.py --cpu-images 0 --live-traffic off 2>/dev/null | show "all legs      "
python bench.py --cpu-images 0 --live-traffic off --from-rgb-steps 0 --latency-reps 0 --lbs-unfused-reps 0 2>/dev/null | show "stress only   "
done
python bench.py --batch 16 --num-samples 1000 --steps 12 --warmup 4 --cpu-images 0 --live-traffic off --from-rgb-steps 0 --latency-reps 0 --lbs-unfused-reps 0 --stress-steps 0 --split-steps 0 --mesh-arith bf16x3 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('own run bf16x3 auto partition:', round(d['value']), d['config']['step_pipelining'][-120:])"
