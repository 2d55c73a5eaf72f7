#!/bin/bash
# Run ON THE GPU BOX (through gpurun): per-dispatch SQ / GRBM counters of one pass over the encoder's launch list
# (tools/encoder_layers.py once), separate rocprofv3 --pmc passes.  Raw CSVs under gpurun_out/encpmc/; merged with the
# stand-alone timings by tools/summarize_encoder_layers.py.
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/encpmc; rm -rf $OUT; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
run() { name=$1; shift; timeout 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT -o $name -- python $R/tools/encoder_layers.py once > $OUT/$name.log 2>&1; echo "$name exit $?"; }
run sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS
run grbm GRBM_GUI_ACTIVE
