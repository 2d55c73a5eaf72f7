#!/bin/bash
# Run ON THE GPU BOX (through gpurun): SQ / GRBM / TCC counters of the fused mesh kernel (hps_smpl_mesh_fused) at 6 528 meshes,
# one rocprofv3 --pmc pass per counter group (FETCH_SIZE and WRITE_SIZE do not fit one pass).  Raw CSVs under
# gpurun_out/meshpmc/; summarised by tools/summarize_profiles.py.
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/meshpmc; rm -rf $OUT; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
K="mesh_fused_kernel"
run() { name=$1; shift; timeout 200 rocprofv3 --pmc "$@" --kernel-trace --kernel-include-regex "$K" --output-format csv -d $OUT -o $name -- python $R/tests/dev/mesh_one.py 6528 fused 6 > $OUT/$name.log 2>&1; echo "$name exit $?"; }
run sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES
run sq2 SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS
run grbm GRBM_GUI_ACTIVE
run fetch FETCH_SIZE
run write WRITE_SIZE
