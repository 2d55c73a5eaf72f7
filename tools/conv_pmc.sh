#!/bin/bash
# Run ON THE GPU BOX (through gpurun): SQ / GRBM counters of one convolution layer under the old (conv.hip v3), its
# no-DMA / no-MFMA ablations and the halo-padded kernel (conv_pad.hip).  Summarised by tools/summarize_conv_pmc.py.
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/convpmc; rm -rf $OUT; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
ARGS="v3:1 v3:21 v3:31 pad:1"
timeout 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --kernel-include-regex "conv_igemm_v3|conv_pad_kernel" --output-format csv -d $OUT -o sq -- python $R/tests/dev/conv_one.py $ARGS > $OUT/sq.log 2>&1; echo "sq exit $?"
timeout 200 rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --kernel-include-regex "conv_igemm_v3|conv_pad_kernel" --output-format csv -d $OUT -o grbm -- python $R/tests/dev/conv_one.py $ARGS > $OUT/grbm.log 2>&1; echo "grbm exit $?"
