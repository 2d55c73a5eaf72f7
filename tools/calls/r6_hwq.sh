cd $GRAFT_REPO_ROOT
show() { python -c "
import json,sys; d=json.loads(sys.stdin.read()); s=d['secondary']; print('$1: headline', round(d['value']), '| bf16x3', round(s['mesh_bf16x3']['images_per_s']), '| from_rgb', [round(x) for x in s['from_rgb']['legs_images_per_s']], '| lat', round(s['latency_b1']['median_ms'],3), round(s['latency_b1']['graph_median_ms'],3), '| stress', round(s['stress_n1000']['images_per_s']), round(s['stress_n1000']['mesh_bf16x3']['images_per_s']))"; }
for rep in 1 2; do
python bench.py --cpu-images 0 --live-traffic off 2>/dev/null | show "default queues"
GPU_MAX_HW_QUEUES=8 python bench.py --cpu-images 0 --live-traffic off 2>/dev/null | show "8 hw queues   "
GPU_MAX_HW_QUEUES=16 python bench.py --cpu-images 0 --live-traffic off 2>/dev/null | show "16 hw queues  "
GPU_MAX_HW_QUEUES=2 python bench.py --cpu-images 0 --live-traffic off 2>/dev/null | show "2 hw queues   "
done
