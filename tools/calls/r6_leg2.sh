cd $GRAFT_REPO_ROOT
show() { python -c "
import json,sys; d=json.loads(sys.stdin.read()); s=d['secondary']['stress_n1000']; print('$1: stress f32', round(s['images_per_s']), round(s['ms_per_step'],3), 'mesh', round(s['mesh_kernel']['median_ms'],3), '| bf16x3', s['mesh_bf16x3'] and (round(s['mesh_bf16x3']['images_per_s']), round(s['mesh_bf16x3']['ms_per_step'],3), round(s['mesh_bf16x3']['mesh_kernel_median_ms'],3)))"; }
for rep in 1 2 3; do
python bench.py --cpu-images 0 --live-traffic off 2>/dev/null | show "all legs      "
python bench.py --cpu-images 0 --live-traffic off --from-rgb-steps 0 --latency-reps 0 --lbs-unfused-reps 0 2>/dev/null | show "stress only   "
python bench.py --cpu-images 0 --live-traffic off --latency-reps 0 2>/dev/null | show "no latency leg"
python bench.py --cpu-images 0 --live-traffic off --from-rgb-steps 0 2>/dev/null | show "no from_rgb   "
done
