R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/prof_hip; rm -rf $OUT; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --hip-runtime-trace --output-format csv -d $OUT -o bench -- python $R/bench.py --steps 10 --warmup 2 --cpu-images 0 --from-rgb-steps 0 --latency-reps 0 --stress-steps 0 --lbs-unfused-reps 0 $QS_EXTRA > $OUT/bench.log 2>&1
echo "exit $?"; ls -la $OUT | head; grep -h '^{' $OUT/bench.log | tail -1 | cut -c1-200
python3 - <<'PY'
import csv, os, glob
d = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/prof_hip/"
api = list(csv.DictReader(open(glob.glob(d + "*hip_api_trace.csv")[0])))
ker = list(csv.DictReader(open(glob.glob(d + "*kernel_trace.csv")[0])))
print(api[0].keys()); print(ker[0].keys())
# keep only launches: small file
launch = {r["Correlation_Id"]: r for r in api if "Launch" in r["Function"]}
with open(d + "launch_vs_start.csv", "w") as f:
    f.write("kernel,queue,host_launch_ns,gpu_start_ns,gpu_end_ns\n")
    for k in ker:
        a = launch.get(k["Correlation_Id"])
        f.write("%s,%s,%s,%s,%s\n" % (k["Kernel_Name"].split("(")[0].replace(",", ";")[:60], k["Queue_Id"], a["Start_Timestamp"] if a else "", k["Start_Timestamp"], k["End_Timestamp"]))
for p in glob.glob(d + "*hip_api_trace.csv"): os.remove(p)       # large
PY
