R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/convpmc; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
timeout 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VMEM --kernel-trace --kernel-include-regex "conv_igemm_v3" --output-format csv -d $OUT -o sq -- python $R/tests/dev/conv_one.py 1 21 31 41 > $OUT/sq.log 2>&1; echo "sq exit $?"
timeout 200 rocprofv3 --pmc GRBM_GUI_ACTIVE GRBM_COUNT --kernel-trace --kernel-include-regex "conv_igemm_v3" --output-format csv -d $OUT -o grbm -- python $R/tests/dev/conv_one.py 1 21 31 41 > $OUT/grbm.log 2>&1; echo "grbm exit $?"
ls $OUT
