#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
OUT=$R/gpurun_out/r6_call5; rm -rf $OUT; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_e2e.py tests/test_gpu_smpl.py -m gpu -x -q > $OUT/pytest.txt 2>&1; echo "pytest exit $?"; tail -12 $OUT/pytest.txt
Q="--cpu-images 0 --from-rgb-steps 0 --stress-steps 0 --latency-reps 0 --lbs-unfused-reps 0 --steps 40 --warmup 10"
{
for rep in 1 2 3; do
  for v in "" "--separate-joints"; do
    python bench.py $Q $v 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('[%-20s] %6d images/s  %.3f ms/step  encoder %.3f mesh %.4f' % ('$v', d['value'], d['ms_per_step'], d['secondary']['encoder']['avg_ms'], d['roofline']['avg_launch_ms']))"
  done
done
} > $OUT/joints_ab.txt 2>&1
cat $OUT/joints_ab.txt
