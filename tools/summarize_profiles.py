"""Turn gpurun_out/prof_<tag>/ (tools/collect_profiles.sh), gpurun_out/meshpmc/ (tools/mesh_pmc.sh) and the encoder table
(tools/summarize_encoder_layers.py) into the committed summaries under profiles/:

  <tag>_bench_line.json / _bench_n1000.json / _bench_unfused.json / _bench_nopipe.json   bench.py JSON lines
  <tag>_bench_kernel_stats.csv, <tag>_n1000_kernel_stats.csv                             rocprofv3 --kernel-trace --stats
  <tag>_mesh_fused_pmc.json (= mesh_fused_pmc_latest.json, quoted by bench.py as roofline.traffic)
  <tag>_kernel_roofline.json / .md    one row per product kernel: algorithmic bytes or FLOPs per launch, time, fraction of peak
"""
import collections
import csv
import json
import os
import shutil
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", "prof_" + tag)
dst = os.path.join(root, "profiles")
os.makedirs(dst, exist_ok=True)
HBM, MFMA = 8000.0, 157.3


def copy(name, to):
    p = os.path.join(src, name)
    if os.path.exists(p) and os.path.getsize(p) > 0:
        shutil.copy(p, os.path.join(dst, to))
        return True
    return False


def load_json(name):
    p = os.path.join(src, name)
    try:
        return json.load(open(p))
    except (OSError, ValueError):
        return None


copy("bench_kernel_stats.csv", tag + "_bench_kernel_stats.csv")
copy("n1000_kernel_stats.csv", tag + "_n1000_kernel_stats.csv")
for a, b in (("bench_line.json", "_bench_line.json"), ("bench_n1000.json", "_bench_n1000.json"),
             ("bench_unfused.json", "_bench_unfused.json"), ("bench_nopipe.json", "_bench_nopipe.json"),
             ("bench_bf16x3.json", "_bench_bf16x3.json"), ("bench_n1000_bf16x3.json", "_bench_n1000_bf16x3.json")):
    copy(a, tag + b)


def pmc_avg(fname, counter, skip=1):
    p = os.path.join(src, fname)
    if not os.path.exists(p):
        return None, 0
    vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(p)) if r["Counter_Name"] == counter]
    vals = vals[skip:] if len(vals) > skip else vals
    return (sum(vals) / len(vals) if vals else None), len(vals)


line = load_json("bench_under_rocprof.json") or load_json("bench_line.json") or {}
M = line.get("config", {}).get("meshes_per_step_per_gpu", 6528)
B = line.get("config", {}).get("images_per_gpu", 64)
N = line.get("config", {}).get("num_samples", 100)

# ---- the mesh kernel's traffic inside the bench (FETCH doubled: MI355X_MICROARCH.md, HBM section) ----
out = {"kernel": "hps::mesh_fused_kernel", "meshes_per_launch": M,
       "command": "rocprofv3 --pmc {FETCH_SIZE|WRITE_SIZE} --kernel-trace --kernel-include-regex mesh_fused_kernel -- "
                  "python bench.py --steps 3 --warmup 1 --cpu-images 0 (separate passes)"}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    v, n = pmc_avg("mesh_%s_counter_collection.csv" % c, c)
    out[c + "_KB_per_launch"], out[c + "_launches"] = v, n
    copy("mesh_%s_counter_collection.csv" % c, "%s_mesh_fused_pmc_%s.csv" % (tag, c))
if out["FETCH_SIZE_KB_per_launch"] is not None and out["WRITE_SIZE_KB_per_launch"] is not None:
    out["hbm_bytes_per_launch"] = (2 * out["FETCH_SIZE_KB_per_launch"] + out["WRITE_SIZE_KB_per_launch"]) * 1024
    out["note"] = ("fabric-side bytes (2 x FETCH_SIZE + WRITE_SIZE, the gfx950 correction of MI355X_MICROARCH.md).  The write is the "
                   "verts output (%.0f MB algorithmic); the fetch is the panel-permuted blend matrix streamed from the 256 MB "
                   "Infinity Cache into the 8 L2s (18.6 MB each pass, counted although it never touches HBM), xt and A."
                   % (M * 82680 / 1e6))
    json.dump(out, open(os.path.join(dst, tag + "_mesh_fused_pmc.json"), "w"), indent=1)
    json.dump(out, open(os.path.join(dst, "mesh_fused_pmc_latest.json"), "w"), indent=1)
print(json.dumps(out, indent=1))

# ---- one roofline row per product kernel, from the kernel stats of the profiled bench ----
stats_p = os.path.join(src, "bench_kernel_stats.csv")
rows = []
if os.path.exists(stats_p):
    stats = list(csv.DictReader(open(stats_p)))
    tot = sum(float(r["TotalDurationNs"]) for r in stats)
    V, J = 6890, 24
    enc_flop = 6.279e9 * B
    stem_flop = 2.0 * B * 128 * 128 * 64 * 49 * 18
    model = [   # (substring, what, bound, algorithmic quantity per launch, unit)
        ("stem_wino_kernel", "stem 7x7/2 in Winograd form + the 3x3/2 max pool in its epilogue (direct-convolution FLOPs; the MFMA pipe does 81/196 of them)", "mfma", stem_flop, "flop"),
        ("conv_", "the other 19 ResNet-18 convolutions, direct + Winograd F(2x2,3x3) (all launches of a step together; direct-convolution FLOPs)", "mfma", enc_flop - stem_flop, "flop/step"),
        ("mesh_fused_kernel", "blend GEMM + LBS, fused (FLOPs by SURVEY 8(d)'s K = 217 per mesh; the shared-shape form multiplies 207 rows per mesh)", "mfma", 2.0 * 217 * 3 * V * M, "flop"),
        ("mesh_split_kernel", "the same kernel with the pose blend GEMM as six bf16 piece products per product (opt-in bf16x3 arithmetic; launched by bench.py's secondary.mesh_bf16x3 leg only; FLOPs: the fp32-equivalent 2 x 207 x 3 V per mesh)", "mfma", 2.0 * 207 * 3 * V * M, "flop"),
        ("split_bf16x3_kernel", "the mesh operand of a call as bf16 piece planes (bf16x3 leg only)", "hbm", M * 208 * 10.0, "bytes"),
        ("uncertainty_joints_kernel", "per-vertex sample uncertainty + the joint regression of every mesh, one launch (bytes: sample vertices read once + uncertainties + the compact regressor vertices + joints)", "hbm",
         (B * N * V * 12.0 + B * V * 4.0) + M * (198 * 12.0 + 90 * 12.0 + 24 * 12.0), "bytes"),
        ("uncertainty_reg_kernel", "per-vertex sample uncertainty", "hbm", (B * N * V * 12.0 + B * V * 4.0), "bytes"),
        ("::joints_kernel", "90 joints per mesh (CSR rows on the compact regressor vertices)", "hbm", M * (198 * 12.0 + 90 * 12.0 + 24 * 12.0), "bytes"),
        ("v_shaped_kernel", "shape blend once per image (smplx lbs step 1 for the shared-shape mesh kernel)", "hbm", B * 3.0 * V * 4 * 2 + 10.0 * 3 * V * 4, "bytes"),
        ("pose_prep_kernel", "Rodrigues / FK / blend operand", "hbm", M * (24 * 9 * 4.0 + 224 * 4.0 + 24 * 12 * 4.0 + 24 * 3 * 4.0 + 40.0), "bytes"),
        ("nchw_to_padded_nhwc", "input relayout", "hbm", 2.0 * B * 18 * 256 * 256 * 4, "bytes"),
        ("stem_phase_split_kernel", "input -> four phase frames per image", "hbm", 2.0 * B * 18 * 256 * 256 * 4, "bytes"),
        ("maxpool_pad_kernel", "3x3/2 max pool", "hbm", B * 64 * 4.0 * (128 * 128 + 64 * 64), "bytes"),
        ("stem_pool_borders_kernel", "border pass of the max pool formed in the stem's epilogue (15 of 64 pooled pixels per item + the side buffer)", "hbm",
         B * 64 * (2 * 15 * 64 * 4.0 + 2048 * 4.0), "bytes"),
        ("mf_sample_kernel", "matrix-Fisher rejection sampling", "alu", B * 23 * 8.0 * N, "proposals"),
        ("joint_level_kernel", "head: per-level MLPs + in-kernel SVD (8 launches per step)", "latency", None, ""),
        ("linear_kernel", "head: FC trunk (3 launches per step)", "latency", None, ""),
    ]
    # encoder passes in the profiled run (timed steps + warm-up + the unfused-LBS repetitions behind the timed region): one stem (or max-pool) launch each
    steps = float(sum(int(r["Calls"]) for r in stats if "stem_wino_kernel" in r["Name"]) or
                  sum(int(r["Calls"]) for r in stats if "maxpool_pad_kernel" in r["Name"]) or 12)
    for sub, what, bound, qty, unit in model:
        rs = [r for r in stats if sub in r["Name"]]
        if not rs:
            continue
        calls = sum(int(r["Calls"]) for r in rs)
        total_ns = sum(float(r["TotalDurationNs"]) for r in rs)
        row = {"kernel": sub, "what": what, "bound": bound, "calls": calls, "total_ms": total_ns / 1e6,
               "avg_us": total_ns / calls / 1e3, "share_of_kernel_time": total_ns / tot}
        if qty is not None:
            per_launch_s = (total_ns / steps if unit.endswith("/step") else total_ns / calls) * 1e-9
            row["algorithmic_per_launch"], row["unit"] = qty, unit
            if bound == "mfma":
                row["achieved_tflops"] = qty / per_launch_s / 1e12
                row["frac_of_peak"] = row["achieved_tflops"] / MFMA
            elif bound == "hbm":
                row["achieved_gbs"] = qty / per_launch_s / 1e9
                row["frac_of_peak"] = row["achieved_gbs"] / HBM
            elif bound == "alu":
                row["proposals_per_s"] = qty / per_launch_s
        rows.append(row)
    json.dump({"tag": tag, "source": "rocprofv3 --kernel-trace --stats of python bench.py --steps 10 --warmup 2 (12 steps; kernels that run "
               "beside the next batch's encoder are stretched by sharing the GPU)", "peaks": {"hbm_gbs": HBM, "mfma_fp32_tflops": MFMA},
               "kernels": rows}, open(os.path.join(dst, tag + "_kernel_roofline.json"), "w"), indent=1)
    md = ["| kernel | what | bound | calls | avg us | achieved | of peak | share of kernel time |", "|---|---|---|---|---|---|---|---|"]
    for r in rows:
        ach = ("%.1f TF/s" % r["achieved_tflops"]) if "achieved_tflops" in r else ("%.0f GB/s" % r["achieved_gbs"]) if "achieved_gbs" in r \
            else ("%.1f G proposals/s" % (r["proposals_per_s"] / 1e9)) if "proposals_per_s" in r else "-"
        md.append("| %s | %s | %s | %d | %.1f | %s | %s | %.1f %% |" % (r["kernel"], r["what"], r["bound"], r["calls"], r["avg_us"], ach,
                                                                      ("%.2f" % r["frac_of_peak"]) if "frac_of_peak" in r else "-",
                                                                      100 * r["share_of_kernel_time"]))
    open(os.path.join(dst, tag + "_kernel_roofline.md"), "w").write("\n".join(md) + "\n")
    print("\n".join(md))
    print("kernel time per profiled run: %.2f ms" % (tot / 1e6))

# ---- round-4 additions: the driver-flag bench line, batch-1 latency (+ kernel timeline), widened rows, predict loop, mesh-kernel SQ / LDS counters
for a, b in (("bench_driver_flags.json", "_bench_driver_flags.json"), ("latency_b1.txt", "_latency_b1.txt"),
             ("latency_b1_timeline.txt", "_latency_b1_timeline.txt"), ("next_rows.txt", "_next_rows.txt"),
             ("predict_time.txt", "_predict_time.txt"), ("mesh_pmc_lds.txt", "_mesh_pmc_lds.txt"), ("ab.txt", "_ab.txt"),
             ("step_launches.txt", "_step_launches.txt"), ("step_timeline.txt", "_step_timeline.txt"),
             ("next_rows_stats.log", "_next_rows_kernels.txt"), ("bench_nographlat.json", "_bench_nographlat.json"),
             ("mesh_bf16x3.txt", "_mesh_bf16x3.txt")):
    copy(a, tag + b)
