#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
OUT=$R/gpurun_out/r6_tests; rm -rf $OUT; mkdir -p $OUT
timeout 2400 python -m pytest tests -m gpu -x -q > $OUT/pytest.txt 2>&1; echo "pytest exit $?"; tail -8 $OUT/pytest.txt
python __graft_entry__.py smoke 2>&1 | tail -4
timeout 300 python tests/dev/conv_one.py 2>&1 | tail -3
