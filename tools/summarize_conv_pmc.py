"""gpurun_out/convpmc (tools/conv_pmc.sh) -> profiles/<tag>_conv_pmc.json: per kernel, averages over the launches after
the first two.  MFMA pipe utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs)."""
import collections
import csv
import json
import os
import re
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", "convpmc")
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in ("sq", "grbm"):
    for r in csv.DictReader(open(os.path.join(src, f + "_counter_collection.csv"))):
        name = r["Kernel_Name"]
        if "conv_pad_kernel" in name:
            key = "conv_pad_kernel<128,128>"
        else:
            ab = re.search(r"64, (\d)>", name).group(1)
            key = {"0": "conv_igemm_v3<128,128>", "1": "conv_igemm_v3 without its LDS-DMA (ablation)",
                   "2": "conv_igemm_v3 without its MFMAs (ablation)"}.get(ab, "v3 ablate " + ab)
        agg[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
        if r["Counter_Name"] in ("SQ_WAVE_CYCLES", "GRBM_GUI_ACTIVE"):
            agg[key]["duration_us_" + f].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
out = {"layer": "3x3 conv 128->128, 32x32, B=64 (19.3 GFLOP), relu(randn) input", "kernels": {}}
for k, d in agg.items():
    m = {c: sum(v[2:]) / len(v[2:]) for c, v in d.items()}
    if "GRBM_GUI_ACTIVE" in m and "SQ_VALU_MFMA_BUSY_CYCLES" in m:
        m["mfma_pipe_busy_frac"] = m["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * m["GRBM_GUI_ACTIVE"] / 8)
        m["tflops"] = 19.327 / m["duration_us_grbm"] * 1e3
    out["kernels"][k] = {c: round(x, 3) for c, x in m.items()}
json.dump(out, open(os.path.join(root, "profiles", tag + "_conv_pmc.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
