cd $GRAFT_REPO_ROOT
for a in f32 bf16x3; do
python - <<PY
import json, subprocess, sys
out = subprocess.run([sys.executable, "bench.py", "--steps", "2000", "--warmup", "10", "--cpu-images", "0", "--live-traffic", "off", "--from-rgb-steps", "0",
                      "--latency-reps", "0", "--stress-steps", "0", "--lbs-unfused-reps", "0", "--split-steps", "0", "--mesh-arith", "$a"], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL).stdout.decode()
d = json.loads([l for l in out.splitlines() if l.startswith("{")][-1])
ls = d["roofline"]["launch_spread"]
print("$a: 2000 steps (128 000 images): %d images/s, %.3f ms/step, mesh kernel median %.4f min %.4f max %.4f ms, checksum images %d" % (d["value"], d["ms_per_step"], ls["median_ms"], ls["min_ms"], ls["max_ms"], d["metric_checksums"]["images"]))
PY
done
rocm-smi --showmemuse 2>/dev/null | grep -i "GPU\[0\]" | head -3
