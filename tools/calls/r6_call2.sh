#!/bin/bash
# Round 6, call 2: GPU tests on the new defaults (NCHW-fed stem, side kernels in stream order) + A/B of the uncertainty pass's block order
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
OUT=$R/gpurun_out/r6_call2; rm -rf $OUT; mkdir -p $OUT
timeout 2400 python -m pytest tests -m gpu -x -q > $OUT/pytest.txt 2>&1; echo "pytest exit $?"; tail -5 $OUT/pytest.txt
Q="--cpu-images 0 --from-rgb-steps 0 --stress-steps 0 --latency-reps 0 --lbs-unfused-reps 0 --steps 40 --warmup 10"
{
for rep in 1 2 3; do
  for m in 0 6; do
    python tests/dev/ab_bench.py hps_dev_unc_mode $m $Q 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('unc mode $m: %6d images/s  %.3f ms/step  encoder %.3f mesh %.3f' % (d['value'], d['ms_per_step'], d['secondary']['encoder']['avg_ms'], d['roofline']['avg_launch_ms']))"
  done
done
} > $OUT/unc_ab.txt 2>&1
cat $OUT/unc_ab.txt
