"""CPU, world_size 2 over gloo: the multi-GPU path's only collective (one all-gather of the metric
accumulator) and the shard partition, exercised with real processes."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from hierarchicalprobabilistic3dhuman_amd import sharding


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, total, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    r, w, _ = sharding.init_distributed("gloo")
    lo, hi = sharding.shard_range(total, r, w)
    # per-image "metrics" keyed by the global image index, as the Philox sampler is
    per_image = torch.tensor([[1.0, (i * 0.37) % 1.0, float(i)] for i in range(lo, hi)], dtype=torch.float64).reshape(-1, 3)
    local = per_image.sum(0)
    per_rank, total_sums = sharding.gather_metric_sums(local)
    q.put((r, lo, hi, per_rank.tolist(), total_sums.tolist()))
    dist.barrier()
    dist.destroy_process_group()


def test_all_gather_metric_reduction_world2():
    world, total = 2, 13
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, total, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, lo0, hi0, pr0, tot0), (r1, lo1, hi1, pr1, tot1) = results
    assert (lo0, hi0, lo1, hi1) == (0, 7, 7, 13)
    assert pr0 == pr1 and tot0 == tot1                       # every rank holds the same gathered table and total
    # equals the single-process run that adds the same shard sums in rank order (bit for bit)
    shard_sums = []
    for lo, hi in ((0, 7), (7, 13)):
        shard_sums.append(torch.tensor([[1.0, (i * 0.37) % 1.0, float(i)] for i in range(lo, hi)], dtype=torch.float64).sum(0))
    want = torch.zeros(3, dtype=torch.float64)
    for s in shard_sums:
        want = want + s
    assert tot0 == want.tolist() and tot0[0] == total


def test_single_process_is_world_size_one():
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        os.environ.pop(k, None)
    assert sharding.init_distributed() == (0, 1, 0)
    per_rank, total = sharding.gather_metric_sums(torch.tensor([2.0, 3.0]))
    assert per_rank.tolist() == [[2.0, 3.0]] and total.tolist() == [2.0, 3.0]
