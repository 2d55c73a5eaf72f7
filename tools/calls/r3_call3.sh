#!/bin/bash
# round 3, GPU call 3: CU-partition sweep at configs[4]
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r3c3; mkdir -p $O
cd $R
for k in 0 6 8 10 12 16; do
  timeout 300 python bench.py --batch 16 --num-samples 1000 --steps 12 --warmup 3 --cpu-images 0 --lbs-unfused-reps 0 --mesh-overlap on --encoder-cus $k > $O/n1000_k$k.log 2>&1; echo "n1000 k=$k rc $?"
done
timeout 300 python bench.py --batch 16 --num-samples 1000 --steps 12 --warmup 3 --cpu-images 0 --lbs-unfused-reps 0 --mesh-overlap off > $O/n1000_off.log 2>&1
for k in 8 12; do
  timeout 300 python bench.py --steps 20 --warmup 5 --cpu-images 0 --lbs-unfused-reps 0 --mesh-overlap on --encoder-cus $k > $O/b64_k$k.log 2>&1; echo "b64 k=$k rc $?"
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r3c3/*.log')):
    l=[x for x in open(f) if x.startswith('{')]
    if not l: print(f,'NO JSON', open(f).read()[-600:]); continue
    d=json.loads(l[-1]); print(f.split('/')[-1], round(d['value']), '%.3f ms/step'%d['ms_per_step'], 'mesh %.3f'%d['roofline']['avg_launch_ms'], {k:round(v.get('avg_ms',0),3) for k,v in d['secondary'].items()}, d['metric_checksums']['sum_unc'])
PY
