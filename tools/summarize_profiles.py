"""Turn gpurun_out/prof_<tag>/ (tools/collect_profiles.sh) into the committed summaries under profiles/."""
import csv
import json
import os
import shutil
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", "prof_" + tag)
dst = os.path.join(root, "profiles")
os.makedirs(dst, exist_ok=True)
shutil.copy(os.path.join(src, "bench_kernel_stats.csv"), os.path.join(dst, tag + "_bench_kernel_stats.csv"))
out = {"kernel": "hps::lbs_kernel", "command": "rocprofv3 --pmc {FETCH_SIZE|WRITE_SIZE} --kernel-trace "
       "--kernel-include-regex lbs_kernel -- python bench.py --steps 3 --warmup 1 --cpu-images 0 (separate passes)"}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    p = os.path.join(src, "lbs_%s_counter_collection.csv" % c)
    rows = [r for r in csv.DictReader(open(p)) if r["Counter_Name"] == c]
    vals = [float(r["Counter_Value"]) for r in rows]
    out[c + "_KB_per_launch"] = sum(vals) / len(vals)
    out[c + "_launches"] = len(vals)
    shutil.copy(p, os.path.join(dst, "%s_lbs_pmc_%s.csv" % (tag, c)))
# MI355X_MICROARCH.md (HBM): counters are in KB; on gfx950 FETCH_SIZE reports half of a wide coalesced streaming
# read -> doubled; WRITE_SIZE taken as is (it matches the algorithmic write bytes to <1 %)
out["hbm_bytes_per_launch"] = (2 * out["FETCH_SIZE_KB_per_launch"] + out["WRITE_SIZE_KB_per_launch"]) * 1024
stats = {r["Name"]: r for r in csv.DictReader(open(os.path.join(src, "bench_kernel_stats.csv")))}
lbs = [r for n, r in stats.items() if "lbs_kernel" in n]
if lbs:
    out["rocprof_avg_launch_ns"] = float(lbs[0]["AverageNs"])
    out["rocprof_calls"] = int(lbs[0]["Calls"])
try:      # the workload the counters belong to (bench.py only quotes `traffic` for the same launch size)
    line = json.load(open(os.path.join(src, "bench_under_rocprof.json")))
    out["meshes_per_launch"] = line["config"]["meshes_per_step_per_gpu"]
except (OSError, ValueError, KeyError):
    out["meshes_per_launch"] = 6528
json.dump(out, open(os.path.join(dst, tag + "_lbs_pmc.json"), "w"), indent=1)
json.dump(out, open(os.path.join(dst, "lbs_pmc_latest.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
tot = sum(float(r["TotalDurationNs"]) for r in stats.values())
print("kernel time per profiled run: %.2f ms" % (tot / 1e6))
for n, r in sorted(stats.items(), key=lambda kv: -float(kv[1]["TotalDurationNs"]))[:14]:
    print("%-72s calls %5s avg %9.1f us %6s%%" % (n.replace("void ", "")[:72], r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
