cd $GRAFT_REPO_ROOT
for rep in 1 2; do
python bench.py --cpu-images 0 --live-traffic off 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); m=d['secondary']['mesh_bf16x3']; print('default flags: headline', round(d['value']), '| bf16x3 leg', round(m['images_per_s']), round(m['ms_per_step'],3), 'enc', round(m['encoder_avg_ms'],3), 'mesh', round(m['mesh_kernel_ms']['median_ms'],4))"
python bench.py --cpu-images 0 --live-traffic off --stress-steps 0 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); m=d['secondary']['mesh_bf16x3']; print('no stress leg: headline', round(d['value']), '| bf16x3 leg', round(m['images_per_s']), round(m['ms_per_step'],3), 'enc', round(m['encoder_avg_ms'],3), 'mesh', round(m['mesh_kernel_ms']['median_ms'],4))"
python bench.py --cpu-images 0 --live-traffic off --from-rgb-steps 0 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); m=d['secondary']['mesh_bf16x3']; print('no from_rgb leg: headline', round(d['value']), '| bf16x3 leg', round(m['images_per_s']), round(m['ms_per_step'],3), 'enc', round(m['encoder_avg_ms'],3), 'mesh', round(m['mesh_kernel_ms']['median_ms'],4))"
python bench.py --cpu-images 0 --live-traffic off --steps 20 --warmup 5 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); m=d['secondary']['mesh_bf16x3']; print('driver flags: headline', round(d['value']), '| bf16x3 leg', round(m['images_per_s']), round(m['ms_per_step'],3), 'enc', round(m['encoder_avg_ms'],3), 'mesh', round(m['mesh_kernel_ms']['median_ms'],4))"
done
