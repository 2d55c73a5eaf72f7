"""gpurun_out/enc_layers_time.json + gpurun_out/encpmc/*_counter_collection.csv -> profiles/<tag>_encoder_layers.{json,md}:
one row per operation of the encoder's launch list at B = 64: time alone, algorithmic GFLOP, TF/s, fraction of the 157.3 TF/s
fp32 MFMA peak, and from the PMC pass MFMA pipe busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs)."""
import collections
import csv
import json
import os
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
t = json.load(open(os.path.join(root, "gpurun_out", "enc_layers_time.json")))


def per_dispatch(fname):
    """dispatch id -> {counter: value}, kernel name; in dispatch order"""
    d, names = collections.OrderedDict(), {}
    path = os.path.join(root, "gpurun_out", "encpmc", fname)
    if not os.path.exists(path):
        return d, names
    for r in csv.DictReader(open(path)):
        k = int(r["Dispatch_Id"])
        d.setdefault(k, {})[r["Counter_Name"]] = float(r["Counter_Value"])
        names[k] = r["Kernel_Name"]
    return d, names


sq, sq_names = per_dispatch("sq_counter_collection.csv")
grbm, _ = per_dispatch("grbm_counter_collection.csv")
# launches per pass over the list: one per operation, two for split-K convolutions
per_pass = sum(r.get("launches", 1) for r in t["ops"])


def last_pass(d):
    keys = [k for k in sorted(d) if "conv_pad" in sq_names.get(k, "conv_pad") or True]
    return keys[-per_pass:] if len(keys) >= per_pass else []


ks, kg = last_pass(sq), last_pass(grbm)
i = 0
for r in t["ops"]:
    n = r.get("launches", 1)
    if ks and kg:
        c = sq[ks[i]]
        g = grbm[kg[i]]["GRBM_GUI_ACTIVE"]
        r["kernel"] = sq_names[ks[i]].split("(")[0].replace("void ", "")[:60]
        r["mfma_pipe_busy"] = c["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * g / 8.0) if g else None
        r["valu_insts_per_mfma_busy_kcycle"] = c["SQ_INSTS_VALU"] / max(1.0, c["SQ_VALU_MFMA_BUSY_CYCLES"] / 1e3)
        r["sq_wait_inst_any_frac"] = c["SQ_WAIT_INST_ANY"] / max(1.0, c["SQ_WAVE_CYCLES"])
    if "tflops" in r:
        r["frac_of_peak"] = r["tflops"] / 157.3
    i += n
out = dict(t, tag=tag, peak_tflops=157.3,
           how="tools/encoder_layers.py time (HIP events, 20 launches of each operation alone) + tools/encoder_pmc.sh "
               "(rocprofv3 --pmc, one pass over the launch list); MFMA pipe busy measured under the profiler's clocks")
dst = os.path.join(root, "profiles")
json.dump(out, open(os.path.join(dst, tag + "_encoder_layers.json"), "w"), indent=1)
lines = ["| operation | kernel | ms alone (20 back-to-back launches) | ms in the list (events between the operations) | direct-conv GFLOP | TF/s alone (direct-conv equivalent) | of 157.3 | MFMA pipe busy (PMC) |", "|---|---|---|---|---|---|---|---|"]
for r in t["ops"]:
    lines.append("| %s | %s | %.4f | %s | %s | %s | %s | %s |" % (
        r["op"], ("winograd" if "winograd" in r.get("algorithm", "") else "direct") if "gflop" in r else "-",
        r["ms_alone"], "%.4f" % r["ms_in_list"] if "ms_in_list" in r else "-", "%.2f" % r["gflop"] if "gflop" in r else "-", "%.1f" % r["tflops"] if "tflops" in r else "-",
        "%.2f" % r["frac_of_peak"] if "frac_of_peak" in r else "-",
        "%.2f" % r["mfma_pipe_busy"] if r.get("mfma_pipe_busy") is not None and "gflop" in r else "-"))
lines.append("| **whole encoder (one launch list)** | | %.3f | | %.1f | %.1f | %.2f | |" % (
    t["whole_encoder_ms"], 6.279 * t["batch"], t["encoder_tflops"], t["encoder_tflops"] / 157.3))
open(os.path.join(dst, tag + "_encoder_layers.md"), "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
