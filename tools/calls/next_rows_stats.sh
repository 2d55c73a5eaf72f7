R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/prof_rows; rm -rf $OUT; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o rows -- python $R/tools/next_rows_time.py > $OUT/rows.log 2>&1
echo "exit $?"; cat $OUT/rows.log | grep -v amdgpu.ids
python3 - <<'PY'
import csv,os
p=os.environ.get("GRAFT_REPO_ROOT")+"/gpurun_out/prof_rows/rows_kernel_stats.csv"
for r in csv.DictReader(open(p)):
    print("%-100s calls %5s avg %8.2f us min %8.2f max %8.2f" % (r["Name"][:100], r["Calls"], float(r["AverageNs"])/1e3, float(r["MinNs"])/1e3, float(r["MaxNs"])/1e3))
PY
rm -f $OUT/rows_kernel_trace.csv
