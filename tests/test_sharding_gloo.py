"""CPU, world_size 2 over gloo: the multi-GPU path's only collective (one all-gather of the metric
accumulator) and the shard partition, exercised with real processes."""
import os
import socket

import pytest
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
    # the evaluation's run seed: ranks whose generators are seeded differently still end up with rank 0's draw
    torch.manual_seed(100 + r)
    mine = int(torch.randint(0, 2 ** 62, (1,), dtype=torch.int64).item())
    seed = sharding.broadcast_int(mine)
    q.put((r, lo, hi, per_rank.tolist(), total_sums.tolist(), mine, seed))
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
    (r0, lo0, hi0, pr0, tot0, mine0, seed0), (r1, lo1, hi1, pr1, tot1, mine1, seed1) = results
    assert mine0 != mine1 and seed0 == seed1 == mine0        # broadcast_int: rank 0's value everywhere
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


def _eval_worker(rank, world, port, n_frames, tmp, q):
    """The evaluation harness's multi-rank logic without a GPU: dataset sharding, the tracker's one collective, the
    per-frame gather (metric values are synthetic: the device kernels are not under test here)."""
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    sharding.init_distributed("gloo")
    from hierarchicalprobabilistic3dhuman_amd.eval_metrics_tracker import EvalMetricsTracker
    frames = list(range(n_frames))
    shard = sharding.shard_dataset(frames)
    mine = [shard[i] for i in range(len(shard))]
    tracker = EvalMetricsTracker(["PVE", "MPJPE"], save_path=tmp, save_per_frame_metrics=True)
    tracker.initialise_metric_sums()
    tracker.initialise_per_frame_metric_lists()
    for f in mine:                                    # what update_per_batch accumulates, with made-up per-frame errors
        tracker.total_samples += 1
        tracker.metric_sums["PVE"] = tracker.metric_sums["PVE"] + torch.tensor(6890.0 * (f + 1), dtype=torch.float64)
        tracker.metric_sums["MPJPE"] = tracker.metric_sums["MPJPE"] + torch.tensor(14.0 * (0.5 * f), dtype=torch.float64)
        tracker.per_frame_metrics["PVE"].append(torch.tensor([float(f + 1)]))
        tracker.per_frame_metrics["MPJPE"].append(torch.tensor([0.5 * f]))
    tracker.reduce_across_ranks()
    final = tracker.compute_final_metrics(verbose=False)
    names = sharding.gather_per_frame([("frame%03d" % f) for f in mine])
    q.put((rank, mine, tracker.total_samples, final, names.tolist(),
           torch.cat(tracker.per_frame_metrics["PVE"]).tolist()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4, 8])
def test_evaluation_sharding_and_reduction(world, tmp_path):
    """SURVEY 8(e) for the evaluate harness: every frame is evaluated by exactly one rank (contiguous blocks), the reduced
    means equal the single-process means, per-frame records come back in dataset order, rank 0 alone writes them."""
    n_frames = 21
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_eval_worker, args=(r, world, port, n_frames, str(tmp_path), q)) for r in range(world)]
    for p in procs:
        p.start()
    results = sorted(q.get(timeout=240) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    seen = [f for r in results for f in r[1]]
    assert seen == list(range(n_frames))                                   # disjoint, complete, rank order = dataset order
    want_pve = sum(f + 1 for f in range(n_frames)) / n_frames
    want_mpjpe = sum(0.5 * f for f in range(n_frames)) / n_frames
    for rank, mine, total, final, names, per_frame in results:
        assert total == n_frames
        assert abs(final["PVE"] - want_pve) < 1e-12 and abs(final["MPJPE"] - want_mpjpe) < 1e-12
        assert names == ["frame%03d" % f for f in range(n_frames)]
        assert per_frame == [float(f + 1) for f in range(n_frames)]
    import numpy as np
    saved = np.load(os.path.join(str(tmp_path), "PVE_per_frame.npy"))
    assert saved.tolist() == [float(f + 1) for f in range(n_frames)]


def _handover_worker(rank, world, port, q):
    import time
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    sharding.init_distributed("gloo")
    t0 = time.perf_counter()
    if rank == 0:
        time.sleep(1.5)                               # rank 0's long host-side work (bench.py: the CPU baseline)
    sharding.wait_for_rank0("test_handover", timeout_s=60)
    q.put((rank, time.perf_counter() - t0))
    dist.destroy_process_group()


def test_ranks_wait_for_rank0s_host_work_without_a_collective():
    """bench.py with N > 1: rank 0 times the CPU baseline after the run's one collective; the other ranks block on the store
    (sharding.wait_for_rank0) until the line is printed, then everybody tears the group down."""
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_handover_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert results[1] >= 1.0                          # rank 1 really waited for rank 0's signal
    sharding.wait_for_rank0("no_group")               # no process group: a no-op


def test_shard_dataset_partitions_exactly():
    for world in (1, 2, 3, 4, 8):
        for n in (0, 1, 7, 8, 64, 513):
            blocks = [sharding.shard_dataset(list(range(n)), r, world) for r in range(world)]
            flat = [b[i] for b in blocks for i in range(len(b))]
            assert flat == list(range(n))
            assert max(len(b) for b in blocks) - min(len(b) for b in blocks) <= 1


def test_single_process_is_world_size_one():
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        os.environ.pop(k, None)
    assert sharding.init_distributed() == (0, 1, 0)
    per_rank, total = sharding.gather_metric_sums(torch.tensor([2.0, 3.0]))
    assert per_rank.tolist() == [[2.0, 3.0]] and total.tolist() == [2.0, 3.0]
