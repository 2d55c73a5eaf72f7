#!/bin/bash
# run the MKL-rounding searches on the GPU box's host CPU
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R/tools/svd_pin
echo "== geqrf probe"; timeout 600 python qr_probe.py 2>&1 | tail -30
echo "== gebd2 search"; timeout 900 python gebd2_search2.py 2>&1 | tail -12
echo "== rotations (bidiagonal inputs)"; timeout 900 python run_search.py 2>&1 | tail -26
