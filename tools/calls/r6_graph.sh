#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
OUT=$R/gpurun_out/r6_graph; rm -rf $OUT; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_e2e.py tests/test_gpu_sampler.py -m gpu -x -q -k "graph or predict_loop or sampler or sampling" > $OUT/pytest.txt 2>&1; echo "pytest exit $?"; tail -25 $OUT/pytest.txt
for a in "1024 1 50" "1024 1 50 --latency" "1024 2 50 --latency" "2048 64 50"; do timeout 200 python tools/predict_time.py $a 2>&1 | tail -1; done
