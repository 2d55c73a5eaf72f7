#!/bin/bash
# the multi-rank plumbing lines on the one-GPU box: 2 ranks over gloo sharing the device, 1 rank under torch.distributed.run with
# nccl, and the plain run on the same box -> gpurun_out/multirank/*.json (copied to profiles/<tag>_multirank_bench_*.json)
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/multirank; mkdir -p $O; cd $R
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 2 --backend gloo --steps 20 --warmup 5 --cpu-images 0 > $O/2rank_gloo.log 2>&1; echo "2-rank gloo rc $?"
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29612 bench.py --gpus 1 --backend nccl --steps 20 --warmup 5 --cpu-images 0 > $O/1rank_nccl.log 2>&1; echo "1-rank nccl rc $?"
timeout 300 python bench.py --steps 20 --warmup 5 --cpu-images 0 > $O/plain.log 2>&1; echo "plain rc $?"
for f in 2rank_gloo 1rank_nccl plain; do grep -h '^{' $O/$f.log | tail -1 > $O/$f.json; python -c "
import json; d=json.load(open('$O/$f.json')); print('$f', round(d['value']), round(d['ms_per_step'],3), d['n_gpus'], d['backend'])"; done
