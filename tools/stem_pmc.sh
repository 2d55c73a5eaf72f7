#!/bin/bash
# Run ON THE GPU BOX (through gpurun): SQ counters of the Winograd stem kernel (tests/dev/stem_wino_ablate.py with one mode),
# separate rocprofv3 --pmc passes; prints per-dispatch averages.
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/stempmc; rm -rf $OUT; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
export PYTHONPATH=$R
MODE=${1:-0}
run() { name=$1; shift; timeout 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT -o $name -- python $R/tests/dev/stem_wino_ablate.py $MODE > $OUT/$name.log 2>&1; echo "$name exit $?"; }
run a SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_INST_LDS
run b SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_ACTIVE_INST_MISC
run c GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU
python - <<'PY'
import csv, glob, os, collections
out = os.environ.get("GRAFT_REPO_ROOT", os.getcwd()) + "/gpurun_out/stempmc"
for f in sorted(glob.glob(out + "/**/*counter_collection.csv", recursive=True)):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "stem_wino_kernel" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in sorted(acc.items()):
        print("%-32s n=%3d  mean %.4g" % (k, len(v), sum(v) / len(v)))
PY
