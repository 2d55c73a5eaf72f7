#!/bin/bash
# Run ON THE GPU BOX: SQ instruction / LDS counters of the fused mesh kernel at 6 528 meshes (what bounds the
# skinning epilogue) -- two rocprofv3 --pmc passes, raw CSVs under gpurun_out/meshpmc_lds/, printed summary per launch.
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/meshpmc_lds; rm -rf $OUT; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
K="mesh_fused_kernel"
run() { name=$1; shift; timeout 200 rocprofv3 --pmc "$@" --kernel-trace --kernel-include-regex "$K" --output-format csv -d $OUT -o $name -- python $R/tests/dev/mesh_one.py 6528 fused 4 > $OUT/$name.log 2>&1; echo "$name exit $?"; }
run sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES
run sq2 SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS
run sq3 SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY SQ_ACTIVE_INST_MISC SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL
python3 - <<'PY'
import csv, glob, os, collections
out = os.environ.get("GRAFT_REPO_ROOT", ".") + "/gpurun_out/meshpmc_lds"
tot = collections.defaultdict(list)
for f in glob.glob(out + "/*counter_collection.csv"):
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(f)):
        per[r["Dispatch_Id"]][r["Counter_Name"]] += float(r["Counter_Value"])
    for d, cs in per.items():
        for c, v in cs.items():
            tot[c].append(v)
for c in sorted(tot):
    v = sorted(tot[c]); print("%-32s median per launch %.4g  (%d launches)" % (c, v[len(v) // 2], len(v)))
PY
