"""Image sharding over the GPUs of one node and the single metric reduction.

Every image is independent through the whole path (eval-mode BatchNorm, per-image sampling, per-mesh
SMPL), so rank r of R simply owns the contiguous block [r*B/R, (r+1)*B/R) of the global batch; weights and
SMPL constants are replicated.  Nothing on the data path is exchanged.  The only collective is one
all-gather of each rank's small metric accumulator at the end of a run (RCCL over xGMI when the backend is
"nccl"; latency-bound, O(100) bytes), reduced in rank order so the result does not depend on arrival order.
The Philox sampler is keyed by the global image index, so per-image outputs are identical for any R.
"""
import os

import torch
import torch.distributed as dist
import torch.utils.data


def init_distributed(backend=None):
    """Join the process group described by RANK / WORLD_SIZE / MASTER_* (torch.distributed.run sets them).
    Returns (rank, world_size, local_rank).  A single process without those variables is world size 1 and no
    process group is created; under torch.distributed.run a group is created even for one rank, so the same
    collectives run for every N."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    launched = "RANK" in os.environ and "WORLD_SIZE" in os.environ        # torch.distributed.run / torchrun
    if (world > 1 or launched) and not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if torch.cuda.is_available():
            torch.cuda.set_device(local_device(local_rank))
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local_rank


def local_device(local_rank):
    """HIP device index of a local rank: ``local_rank`` modulo the visible device count, so that more ranks than GPUs (the
    world-size-2 functional runs on a one-GPU box over gloo) share devices instead of failing in set_device.  RCCL itself
    refuses two ranks on one device, so the wrap-around is only ever exercised with the gloo backend."""
    n = torch.cuda.device_count() if torch.cuda.is_available() else 0
    return local_rank % n if n else 0


def _host_staged():
    """True when the process group cannot carry device tensors (gloo): collectives then go through host copies.  The
    payloads are O(100) bytes, so the staging copy is irrelevant."""
    return dist.get_backend() != "nccl"


def barrier():
    """Process-group barrier (no-op without a group).  Under nccl (= RCCL) it runs on this rank's current device."""
    if not (dist.is_available() and dist.is_initialized()):
        return
    if dist.get_backend() == "nccl":
        dist.barrier(device_ids=[torch.cuda.current_device()])
    else:
        dist.barrier()


def wait_for_rank0(key, timeout_s=1800):
    """Rank 0 signals ``key`` through the process group's key-value store; every other rank BLOCKS on the store until it appears
    (or ``timeout_s`` passes) -- a hand-over for long host-side work on rank 0 (bench.py: the CPU baseline, 10-30 s on all host
    threads) that must not race the ranks' teardown.  Not a collective: the waiting ranks neither spin a host thread nor occupy a
    device queue, and no collective timeout applies.  No-op without a process group."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return
    import datetime
    try:
        store = dist.distributed_c10d._get_default_store()
    except Exception:                     # a build without the accessor: fall back to the collective
        barrier()
        return
    if dist.get_rank() == 0:
        store.set("hps/" + key, "1")
    else:
        store.wait(["hps/" + key], datetime.timedelta(seconds=timeout_s))


def all_reduce_max(value):
    """MAX over ranks of one Python float (bench.py: the slowest rank's wall time)."""
    if not (dist.is_available() and dist.is_initialized()):
        return float(value)
    dev = torch.device("cpu") if _host_staged() else torch.device("cuda", torch.cuda.current_device())
    t = torch.tensor([float(value)], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def broadcast_int(value, src=0):
    """Rank ``src``'s Python int on every rank (the evaluation's run seed); the value itself without a process group."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return int(value)
    dev = torch.device("cpu") if _host_staged() else torch.device("cuda", torch.cuda.current_device())
    t = torch.tensor([int(value)], dtype=torch.int64, device=dev)
    dist.broadcast(t, src=src)
    return int(t.item())


def shard_range(total, rank, world):
    """Contiguous block of ``total`` items owned by ``rank``; the first ``total % world`` ranks get one extra."""
    base, rem = divmod(total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def gather_metric_sums(local_sums):
    """One all-gather of a rank's 1-D accumulator (sums and a count).  Returns (per_rank (R,K), total (K,)),
    the total being the rank-ordered sum -- bit-identical on every rank and to a single-process run that adds
    the same per-shard sums in the same order."""
    local_sums = local_sums.reshape(-1)
    if not (dist.is_available() and dist.is_initialized()):
        per_rank = local_sums[None].clone()
    else:
        send = local_sums.contiguous()
        if send.is_cuda and _host_staged():
            send = send.cpu()                       # gloo: host copy of a few doubles
        bufs = [torch.empty_like(send) for _ in range(dist.get_world_size())]
        dist.all_gather(bufs, send)
        per_rank = torch.stack(bufs).to(local_sums.device)
    total = torch.zeros_like(local_sums)
    for r in range(per_rank.shape[0]):
        total = total + per_rank[r]
    return per_rank, total


def world_info():
    """(rank, world size) of the initialised process group, (0, 1) without one."""
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def shard_dataset(dataset, rank=None, world=None):
    """The contiguous block of ``dataset`` this rank evaluates (evaluate/evaluate_poseMF_shapeGaussian_net.py:35-40 builds ONE
    loader over the whole set; with R ranks each rank must see 1/R of the frames, or the all-gathered metric sums would
    count every frame R times).  Returns the dataset itself for world size 1."""
    if rank is None or world is None:
        rank, world = world_info()
    if world == 1:
        return dataset
    lo, hi = shard_range(len(dataset), rank, world)
    return torch.utils.data.Subset(dataset, range(lo, hi))


def gather_per_frame(local_array):
    """Per-frame records (numpy array, first axis = this rank's frames, in dataset order) of every rank concatenated in rank
    order = dataset order, on every rank.  A few KB per rank (optional per-frame gather)."""
    import numpy as np
    local_array = np.asarray(local_array)
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return local_array
    parts = [None] * dist.get_world_size()
    dist.all_gather_object(parts, local_array)
    parts = [p for p in parts if p is not None and len(p)]
    return np.concatenate(parts, axis=0) if parts else local_array


_SUM_WS = {}


def effective_cpus():
    """Hardware threads this process may actually use: the affinity mask capped by the cgroup CPU quota (a container
    can report 256 CPUs and be throttled to 16 -- oversubscribing a quota makes CPU work slower, not faster)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path, parse in (("/sys/fs/cgroup/cpu.max", lambda t: t.split()),
                        ("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", None)):
        try:
            with open(path) as f:
                text = f.read().strip()
            if parse is not None:
                quota, period = parse(text)
            else:
                quota = text
                with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
                    period = f.read().strip()
            if quota not in ("max", "-1"):
                n = min(n, max(1, int(float(quota) / float(period) + 0.5)))
            break
        except (OSError, ValueError):
            continue
    return max(1, n)


def batch_metric_sums(result, accumulate=None):
    """Accumulator of one ``infer`` result: [image count, sum of per-vertex uncertainty, sum |mode vertices|,
    sum |sample joints|] in float64 -- the checksum-of-checksums the scaling tests compare across world sizes.
    Two small launches (hps_sums_f64: fixed summation order, no float64 temporaries) -- this runs inside bench.py's timed step.
    ``accumulate``: optional (4,) float64 device tensor, a loop's running total: the second launch adds this batch's sums to it
    (the same values ``accumulate.add_(returned)`` would leave, without that launch)."""
    import ctypes
    from . import _capi
    ts = [_capi.f32c(result["unc"]), _capi.f32c(result["verts_mode"]), _capi.f32c(result["joints_samples"])]
    dev = ts[0].device
    key = (dev, _capi.stream().value)                  # the partial sums live between the two launches: one buffer per stream
    ws = _SUM_WS.get(key)
    if ws is None:
        ws = _SUM_WS[key] = torch.empty(4 * 128, dtype=torch.float64, device=dev)
    out = torch.empty(4, dtype=torch.float64, device=dev)
    xs = (ctypes.c_void_p * 3)(*[t.data_ptr() for t in ts])
    ns = (ctypes.c_int64 * 3)(*[t.numel() for t in ts])
    ab = (ctypes.c_int32 * 3)(0, 1, 1)
    _capi.call("hps_sums_f64", xs, ns, ab, 3, float(ts[0].shape[0]), _capi.ptr(ws, torch.float64), _capi.ptr(out, torch.float64),
               _capi.ptr(accumulate, torch.float64) if accumulate is not None else None, _capi.stream())
    return out
