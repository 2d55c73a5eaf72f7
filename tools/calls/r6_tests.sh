#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
OUT=$R/gpurun_out/r6_tests; rm -rf $OUT; mkdir -p $OUT
timeout 2400 python -m pytest tests -m gpu -x -q > $OUT/pytest.txt 2>&1; echo "pytest exit $?"; tail -6 $OUT/pytest.txt
python __graft_entry__.py smoke 2>&1 | tail -3
timeout 300 python tests/dev/conv_one.py 2>&1 | tail -2
timeout 600 python bench.py --steps 20 --warmup 5 > $OUT/bench_driver.json 2> $OUT/bench_driver.err; echo "bench exit $?"; python -c "
import json; d=json.load(open('$OUT/bench_driver.json')); print(round(d['value']), d['ms_per_step'], d['roofline']['frac'], d['roofline_mfma']['frac'], d['cpu_baseline']['value'], sorted(d['secondary'].keys()), d['secondary']['latency_b1'])"
