#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=$R/gpurun_out/ablations_r03_extra.txt; : > $O
sec() { echo; echo "==== $* ===="; }
{
sec "tools/bin/mfma_peak, all modes (the r03 collection above ran the previous binary): random operands prepared OUTSIDE the loop = the sustained fp32 MFMA ceiling"
for i in 1 2; do tools/bin/mfma_peak; done
sec "tools/bin/mfma_lds_overlap (LDS reads / barrier beside 24 or 16 MFMAs per iteration, 1 / 2 / 4 workgroups of 4 waves per CU)"
tools/bin/mfma_lds_overlap | grep -v "SAL=12"
sec "tests/dev/mesh_occ.py (fused mesh kernel at 4 / 3 / 2 / 1 workgroups per CU)"
python tests/dev/mesh_occ.py 2>&1 | grep "^mesh"
sec "tools/calls/trace_step.sh (rocprofv3 kernel trace of the pipelined loop: one steady-state step, every kernel >= 4 us with what overlapped it)"
bash tools/calls/trace_step.sh 2>&1 | grep -v "^exit"
} >> $O 2>&1
echo done
