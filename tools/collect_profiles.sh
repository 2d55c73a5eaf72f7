#!/bin/bash
# Run ON THE GPU BOX (through gpurun): rocprofv3 kernel stats of the default bench command and the separate
# PMC passes for the LBS kernel's HBM traffic.  Writes raw CSVs under gpurun_out/prof_$TAG/.
# usage: tools/collect_profiles.sh r01
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o bench -- python $R/bench.py --steps 10 --warmup 2 --cpu-images 0 > $OUT/bench_stats.log 2>&1
echo "kernel stats exit $?"
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $c --kernel-trace --kernel-include-regex "lbs_kernel" --output-format csv -d $OUT -o lbs_$c -- python $R/bench.py --steps 3 --warmup 1 --cpu-images 0 > $OUT/pmc_$c.log 2>&1
  echo "pmc $c exit $?"
done
grep -h '^{' $OUT/bench_stats.log | tail -1 > $OUT/bench_under_rocprof.json
