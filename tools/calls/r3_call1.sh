#!/bin/bash
# round 3, GPU call 1: tests (incl. the new multi-rank ones), bench lines, multi-rank logs, MFMA microbenchmarks
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r3c1; mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -5 $O/pytest.log
timeout 300 python bench.py --steps 20 --warmup 5 > $O/bench_default.log 2>&1; echo "bench rc $?"; grep -h '^{' $O/bench_default.log | tail -1 > $O/bench_default.json
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 2 --backend gloo --steps 20 --warmup 5 > $O/bench_2rank_gloo.log 2>&1; echo "2-rank gloo rc $?"
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29612 bench.py --gpus 1 --backend nccl --steps 20 --warmup 5 --cpu-images 0 > $O/bench_1rank_nccl.log 2>&1; echo "1-rank nccl rc $?"
timeout 300 python bench.py --steps 20 --warmup 5 --cpu-images 0 > $O/bench_plain.log 2>&1; echo "plain rc $?"
timeout 300 python bench.py --batch 16 --num-samples 1000 --steps 10 --warmup 3 --cpu-images 0 > $O/bench_n1000.log 2>&1; echo "n1000 rc $?"
{ for i in 1 2 3; do timeout 120 tools/bin/mfma_peak; done; for i in 1 2; do timeout 120 tools/bin/mfma_valu_overlap; done; } > $O/mfma_tools.txt 2>&1; echo "tools rc $?"
for f in bench_default bench_2rank_gloo bench_1rank_nccl bench_plain bench_n1000; do echo "== $f"; grep -h '^{' $O/$f.log | tail -1 | cut -c1-400; done
tail -30 $O/mfma_tools.txt
