#!/bin/bash
# Run ON THE GPU BOX: SQ instruction counters and fabric traffic of the row-marching Canny kernel on 64 crops of 3 x 256 x 256 (all
# outputs) -- separate rocprofv3 --pmc passes, raw CSVs under gpurun_out/cannypmc/, printed summary per launch.
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/cannypmc; rm -rf $OUT; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
K="canny_rows_kernel"
run() { name=$1; shift; timeout 200 rocprofv3 --pmc "$@" --kernel-trace --kernel-include-regex "$K" --output-format csv -d $OUT -o $name -- python $R/tests/dev/canny_one.py 64 full 6 > $OUT/$name.log 2>&1; echo "$name exit $?"; }
run sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES GRBM_GUI_ACTIVE
run sq2 SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC
run sq3 SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQC_TC_INST_REQ
run sq4 SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_BRANCH SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD
run fetch FETCH_SIZE
run write WRITE_SIZE
python3 - <<'PY'
import csv, glob, os, collections
out = os.environ.get("GRAFT_REPO_ROOT", ".") + "/gpurun_out/cannypmc"
tot = collections.defaultdict(list)
for f in glob.glob(out + "/*counter_collection.csv"):
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(f)):
        per[r["Dispatch_Id"]][r["Counter_Name"]] += float(r["Counter_Value"])
    for d, cs in per.items():
        for c, v in cs.items():
            tot[c].append(v)
med = {}
for c in sorted(tot):
    v = sorted(tot[c]); med[c] = v[len(v) // 2]; print("%-32s median per launch %.4g  (%d launches)" % (c, med[c], len(v)))
if "FETCH_SIZE" in med and "WRITE_SIZE" in med:
    # MI355X_MICROARCH.md: FETCH_SIZE / WRITE_SIZE in KB, the gfx950 fetch counter reports half the bytes
    b = (2 * med["FETCH_SIZE"] + med["WRITE_SIZE"]) * 1024
    print("fabric bytes per launch (2 x FETCH_SIZE + WRITE_SIZE) %.1f MB; algorithmic 184.5 MB (50.3 in, 134.2 out): ratio %.2f" % (b / 1e6, b / 184.5e6))
if "SQ_INSTS_VALU" in med and "SQ_WAVES" in med:
    w = med["SQ_WAVES"]
    print("per wave: %.0f VALU, %.0f SALU instructions; per row step (24 steps per 16-row strip): %.0f VALU" % (med["SQ_INSTS_VALU"] / w, med["SQ_INSTS_SALU"] / w, med["SQ_INSTS_VALU"] / w / 24))
PY
