#!/bin/bash
# round 3, GPU call 2: mesh/encoder overlap A/B at configs[4] and configs[1]
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r3c2; mkdir -p $O
cd $R
for mo in off on; do
  timeout 300 python bench.py --batch 16 --num-samples 1000 --steps 12 --warmup 3 --cpu-images 0 --lbs-unfused-reps 0 --mesh-overlap $mo > $O/n1000_$mo.log 2>&1; echo "n1000 $mo rc $?"
  timeout 300 python bench.py --steps 20 --warmup 5 --cpu-images 0 --lbs-unfused-reps 0 --mesh-overlap $mo > $O/b64_$mo.log 2>&1; echo "b64 $mo rc $?"
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r3c2/*.log')):
    l=[x for x in open(f) if x.startswith('{')]
    if not l: print(f,'NO JSON'); continue
    d=json.loads(l[-1]); print(f.split('/')[-1], round(d['value']), '%.3f ms/step'%d['ms_per_step'], 'mesh %.3f'%d['roofline']['avg_launch_ms'], {k:round(v.get('avg_ms',0),3) for k,v in d['secondary'].items()})
PY
