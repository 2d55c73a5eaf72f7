R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/rgbtrace; rm -rf $OUT; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d $OUT -o t -- python $R/bench.py --steps 4 --warmup 2 --cpu-images 0 --latency-reps 0 --lbs-unfused-reps 0 --from-rgb-steps 12 > $OUT/log.txt 2>&1
echo exit $?
ls $OUT
python3 - <<'PY'
import csv, os, collections
out=os.environ.get("GRAFT_REPO_ROOT",".")+"/gpurun_out/rgbtrace"
for f in os.listdir(out):
    if f.endswith("kernel_stats.csv"):
        rows=list(csv.DictReader(open(out+"/"+f)))
        for r in rows:
            n=r["Name"]
            if "copy" in n.lower() or "fill" in n.lower() or "rocclr" in n.lower() or "blit" in n.lower(): print("KERNEL", n[:80], r["Calls"], r["TotalDurationNs"])
    if f.endswith("memory_copy_stats.csv"):
        print(open(out+"/"+f).read()[:1500])
PY
