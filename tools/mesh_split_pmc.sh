#!/bin/bash
# Run ON THE GPU BOX: SQ / LDS / HBM counters of the bf16x3 mesh kernel (hps_smpl_mesh_fused_shared_shape_bf16x3) at 6 528 meshes,
# one rocprofv3 --pmc pass per counter group; raw CSVs under gpurun_out/meshsplit_pmc/, printed summary per launch.
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/meshsplit_pmc; rm -rf $OUT; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
K="mesh_split_kernel"
run() { name=$1; shift; timeout 200 rocprofv3 --pmc "$@" --kernel-trace --kernel-include-regex "$K" --output-format csv -d $OUT -o $name -- python $R/tests/dev/mesh_split_time.py --only bf16x3 --reps 4 > $OUT/$name.log 2>&1; echo "$name exit $?"; }
run sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES
run sq2 SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS
run sq3 SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC SQ_LDS_ADDR_CONFLICT
run grbm GRBM_GUI_ACTIVE
run fetch FETCH_SIZE
run write WRITE_SIZE
python3 - <<'PY'
import csv, glob, os, collections
out = os.environ.get("GRAFT_REPO_ROOT", ".") + "/gpurun_out/meshsplit_pmc"
tot = collections.defaultdict(list)
for f in glob.glob(out + "/**/*counter_collection.csv", recursive=True):
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(f)):
        per[r["Dispatch_Id"]][r["Counter_Name"]] += float(r["Counter_Value"])
    for d, cs in per.items():
        for c, v in cs.items():
            tot[c].append(v)
for c in sorted(tot):
    v = sorted(tot[c]); print("%-32s median per launch %.4g  (%d launches)" % (c, v[len(v) // 2], len(v)))
if "FETCH_SIZE" in tot and "WRITE_SIZE" in tot:
    f, w = sorted(tot["FETCH_SIZE"]), sorted(tot["WRITE_SIZE"])
    print("fabric-side bytes per launch = (2 x FETCH_SIZE + WRITE_SIZE) KB = %.1f MB" % ((2 * f[len(f) // 2] + w[len(w) // 2]) * 1024 / 1e6))
PY
