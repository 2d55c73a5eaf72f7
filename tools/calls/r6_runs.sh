cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do
python bench.py --steps 20 --warmup 5 --live-traffic off --cpu-images 0 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); s=d['secondary']; print('driver flags: %6d images/s  %.3f ms/step  encoder %.3f  mesh %.4f | bf16x3 leg %6d | from_rgb %6d | latency_b1 %.3f ms' % (d['value'], d['ms_per_step'], s['encoder']['avg_ms'], d['roofline']['avg_launch_ms'], s['mesh_bf16x3']['images_per_s'], s['from_rgb']['images_per_s'], s['latency_b1']['median_ms']))"
done
