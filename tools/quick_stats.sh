R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/prof_tmp; rm -rf $OUT; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o bench -- python $R/bench.py --steps 10 --warmup 2 --cpu-images 0 --from-rgb-steps 0 --latency-reps 0 --stress-steps 0 --lbs-unfused-reps 0 $QS_EXTRA > $OUT/bench_stats.log 2>&1
echo "exit $?"; grep -h '^{' $OUT/bench_stats.log | tail -1 | cut -c1-200
python3 - <<'PY'
import csv,os
p=os.environ.get("GRAFT_REPO_ROOT")+"/gpurun_out/prof_tmp/bench_kernel_stats.csv"
rows=list(csv.DictReader(open(p)))
for r in rows[:22]:
    print("%-90s calls %5s avg %9.1f us  tot %8.2f ms  %5s%%" % (r["Name"][:90], r["Calls"], float(r["AverageNs"])/1e3, float(r["TotalDurationNs"])/1e6, r["Percentage"]))
PY
