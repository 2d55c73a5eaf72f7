"""GPU: the N > 1 path executed for real on a ONE-GPU box.

RCCL refuses two ranks on one device, so world size 2 runs over gloo with both ranks on cuda:0 (``bench.py --backend gloo``:
the same rank -> shard -> barrier -> all_gather -> all_reduce(MAX) code, collectives host-staged); the nccl (= RCCL) backend
itself is exercised with one rank under torch.distributed.run.  What is asserted is what makes the 8-GPU run correct by
construction: rank r of R does exactly the work ``--as-rank r R`` does alone (its image block, its Philox offsets), the gathered
per-rank accumulators are those, in rank order, and the total is their rank-ordered sum -- bit for bit.

The evaluation harness runs with two ranks (gloo, shared device) on the synthetic dataset and must give the single-process
metrics: every frame evaluated by exactly one rank, samples keyed by the global frame index.
"""
import json
import os
import socket
import subprocess
import sys

import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
COMMON = ["--steps", "2", "--warmup", "1", "--cpu-images", "0", "--lbs-unfused-reps", "0", "--latency-reps", "0", "--from-rgb-steps", "0",
          "--stress-steps", "0", "--live-traffic", "off"]


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _bench(extra, ranks=None, timeout=600):
    cmd = [sys.executable]
    if ranks is not None:
        cmd += ["-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(ranks), "--master-addr", "127.0.0.1",
                "--master-port", str(_free_port())]
    cmd += [os.path.join(ROOT, "bench.py")] + COMMON + extra
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    p = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=timeout)
    out = p.stdout.decode(errors="replace")
    assert p.returncode == 0, "bench failed (%s):\n%s\n%s" % (" ".join(cmd), out[-2000:], p.stderr.decode(errors="replace")[-4000:])
    lines = [l for l in out.splitlines() if l.startswith("{")]
    assert len(lines) == 1, "exactly one JSON line expected from rank 0, got %d" % len(lines)
    return json.loads(lines[0])


@pytest.fixture(scope="module")
def single_rank_shards(dev):
    """What rank 0 and rank 1 of a 2-rank job must compute, each run alone without a process group."""
    return [_bench(["--gpus", "1", "--as-rank", str(r), "2"]) for r in range(2)]


def test_two_rank_gloo_bench_is_the_two_disjoint_shards(dev, single_rank_shards):
    two = _bench(["--gpus", "2", "--backend", "gloo"], ranks=2)
    assert two["n_gpus"] == 2 and two["backend"] == "gloo"
    assert two["config"]["global_batch"] == 128 and two["image_range_rank0"] == [0, 64]
    per_rank = two["metric_checksums_per_rank"]
    assert len(per_rank) == 2
    for r in range(2):
        alone = single_rank_shards[r]
        assert alone["image_range_rank0"] == [64 * r, 64 * r + 64]
        assert per_rank[r] == alone["metric_checksums_per_rank"][0], "rank %d did not do the work of shard %d" % (r, r)
    assert per_rank[0] != per_rank[1]                      # different images: different sums
    want = [a + b for a, b in zip(per_rank[0], per_rank[1])]           # rank-ordered float64 sum
    got = [two["metric_checksums"][k] for k in ("images", "sum_unc", "sum_abs_verts_mode", "sum_abs_joints_samples")]
    assert got == want
    assert got[0] == 2 * 64 * 2                            # ranks x images per step x steps
    assert two["value"] > 0 and abs(two["value"] - 2 * 64 * 2 / (two["ms_per_step"] * 2e-3)) <= 1e-6 * two["value"]


def test_bare_bench_command_starts_its_own_ranks(dev, single_rank_shards):
    """The driver's command is the BARE `python bench.py --gpus N ...` (no torchrun wrapper): bench.py must start its N ranks
    itself and print rank 0's one JSON line.  Two ranks over gloo on the one device: same shards, bit for
    bit, as `--as-rank r 2`; and with the default backend (nccl = RCCL: one device per rank) on a one-GPU box it must refuse
    with one clear sentence and a non-zero status instead of a traceback."""
    # (--cpu-images 2: the N > 1 line carries the CPU baseline too -- rank 0 times the oracle after the collective while rank 1
    # blocks on the store, sharding.wait_for_rank0 -- on a two-image sample to keep the test short)
    two = _bench(["--gpus", "2", "--backend", "gloo", "--cpu-images", "2"])              # ranks=None: plain `python bench.py`
    assert two["n_gpus"] == 2 and two["backend"] == "gloo" and two["config"]["global_batch"] == 128
    assert len(two["metric_checksums_per_rank"]) == 2
    for r in range(2):
        assert two["metric_checksums_per_rank"][r] == single_rank_shards[r]["metric_checksums_per_rank"][0], r
    assert two["metric_checksums"]["images"] == 2 * 64 * 2
    # the schema of every N: metric / unit / scaling, roofline from this run's own HIP events, cpu_baseline timed on this host
    assert two["unit"] == "images/s" and two["scaling"] == "weak" and two["higher_is_better"] is True and two["vs_baseline"] is None
    roof = two["roofline"]
    assert roof["bound"] == "hbm" and roof["launches"] == 2 and roof["achieved"] > 0 and 0 < roof["frac"] < 1 and roof["peak"] == 8000.0
    cpu = two["cpu_baseline"]
    assert cpu["kind"] == "port" and cpu["unit"] == "images/s" and cpu["value"] > 0 and cpu["cores"] >= 1 and "2 images" in cpu["sample"]
    assert two["speedup_vs_cpu_baseline"] == pytest.approx(two["value"] / cpu["value"])
    if torch.cuda.device_count() < 2:
        env = dict(os.environ)
        for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
            env.pop(k, None)
        p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"] + COMMON, cwd=ROOT, env=env,
                           stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
        assert p.returncode != 0 and not p.stdout.strip()
        err = p.stderr.decode(errors="replace")
        assert "one device per rank" in err and "--backend gloo" in err and "Traceback" not in err, err[-2000:]


def test_eight_rank_gloo_bench_is_baseline_config_2(dev):
    """BASELINE configs[2] -- 512 images sharded over 8 ranks, 64 per rank -- executed with all 8 ranks on ONE device over gloo
    (what the 8-GPU node runs over RCCL, minus the xGMI transport): rank r owns images [64 r, 64 r + 64), the first and the last
    rank's accumulators equal what `--as-rank r 8` computes alone bit for bit, the total is the rank-ordered sum of the eight."""
    eight = _bench(["--gpus", "8", "--backend", "gloo"], ranks=8, timeout=1500)
    assert eight["n_gpus"] == 8 and eight["config"]["global_batch"] == 512 and eight["image_range_rank0"] == [0, 64]
    per_rank = eight["metric_checksums_per_rank"]
    assert len(per_rank) == 8 and len({tuple(r) for r in per_rank}) == 8          # eight different shards
    for r in (0, 7):
        alone = _bench(["--gpus", "1", "--as-rank", str(r), "8"])
        assert alone["image_range_rank0"] == [64 * r, 64 * r + 64]
        assert per_rank[r] == alone["metric_checksums_per_rank"][0], r
    want = [0.0] * 4
    for row in per_rank:                                                             # rank order, float64
        want = [a + b for a, b in zip(want, row)]
    got = [eight["metric_checksums"][k] for k in ("images", "sum_unc", "sum_abs_verts_mode", "sum_abs_joints_samples")]
    assert got == want and got[0] == 8 * 64 * 2


def test_one_rank_nccl_bench_under_torchrun(dev, single_rank_shards):
    """RCCL initialisation, barrier, all_gather of the float64 accumulator and all_reduce(MAX) with one rank: the nccl code
    path of the driver's N > 1 launches, and the same numbers as the run without a process group."""
    plain = _bench(["--gpus", "1"])
    one = _bench(["--gpus", "1", "--backend", "nccl"], ranks=1)
    assert one["backend"] == "nccl" and plain["backend"] is None
    assert one["metric_checksums"] == plain["metric_checksums"]
    assert one["metric_checksums_per_rank"] == plain["metric_checksums_per_rank"]
    # world size 1 == shard 0 of 1: the first 64 images, which is also what rank 0 of 2 sees
    assert plain["metric_checksums_per_rank"][0] == single_rank_shards[0]["metric_checksums_per_rank"][0]


# ------------------------------------------------------------------------------------------------------------------
def _eval_setup(dev):
    import copy
    from hierarchicalprobabilistic3dhuman_amd import configs, smpl_data
    from hierarchicalprobabilistic3dhuman_amd.canny_edge_detector import CannyEdgeDetector
    from hierarchicalprobabilistic3dhuman_amd.poseMF_shapeGaussian_net import PoseMFShapeGaussianNet
    from hierarchicalprobabilistic3dhuman_amd.smpl_official import SMPL
    cfg = configs.get_cfg_defaults()
    torch.manual_seed(0)
    net = PoseMFShapeGaussianNet(configs.SMPL_PARENTS, cfg).eval().to(dev)
    smpl = SMPL(smpl_data.synthetic_smpl_model(0)).to(dev)
    male = SMPL(smpl_data.synthetic_smpl_model(1), gender="male").to(dev)
    female = SMPL(smpl_data.synthetic_smpl_model(2), gender="female").to(dev)
    det = CannyEdgeDetector(cfg.DATA.EDGE_NMS, cfg.DATA.EDGE_GAUSSIAN_STD, cfg.DATA.EDGE_GAUSSIAN_SIZE, cfg.DATA.EDGE_THRESHOLD).to(dev)
    return cfg, net, smpl, male, female, det


def _evaluate(dev, save_path, batch_size, n_frames=5, num_samples=4):
    from hierarchicalprobabilistic3dhuman_amd.evaluate_poseMF_shapeGaussian_net import evaluate_pose_MF_shapeGaussian_net
    from metric_scenario import METRICS
    from test_gpu_evaluate import _SyntheticEvalDataset
    cfg, net, smpl, male, female, det = _eval_setup(dev)
    ds = _SyntheticEvalDataset(n_frames, wh=256)
    torch.manual_seed(33)                         # the run seed is drawn from the global CPU generator: same on every rank
    return evaluate_pose_MF_shapeGaussian_net(net, cfg, smpl, male, female, det, dev, ds, METRICS, save_path, num_workers=0,
                                              pin_memory=False, save_per_frame_metrics=True, num_samples_for_metrics=num_samples,
                                              sample_on_cpu=False, batch_size=batch_size)


def _eval_worker(rank, world, port, tmp, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch.distributed as dist
    from hierarchicalprobabilistic3dhuman_amd import sharding
    sharding.init_distributed("gloo")
    dev = torch.device("cuda", sharding.local_device(rank))
    final = _evaluate(dev, tmp, batch_size=2)
    q.put((rank, final))
    sharding.barrier()
    dist.destroy_process_group()


def test_two_rank_evaluation_equals_single_process(dev, tmp_path):
    import numpy as np
    from metric_scenario import METRICS
    one_dir, two_dir = tmp_path / "one", tmp_path / "two"
    one_dir.mkdir(); two_dir.mkdir()
    want = _evaluate(dev, str(one_dir), batch_size=1)
    want_b = _evaluate(dev, None, batch_size=5)                     # batch size must not matter either (samples keyed per frame)
    for m in METRICS:
        assert abs(want_b[m] - want[m]) <= 1e-5 * abs(want[m]), m
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_eval_worker, args=(r, world, port, str(two_dir), q)) for r in range(world)]
    for p in procs:
        p.start()
    results = sorted(q.get(timeout=600) for _ in range(world))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    for rank, final in results:
        for m in METRICS:
            # 5 frames over 2 ranks (3 + 2); float64 sums in another association + fp32 kernels at another batch size
            assert abs(final[m] - want[m]) <= 1e-5 * abs(want[m]), (rank, m, final[m], want[m])
    assert results[0][1] == results[1][1]                           # every rank holds the same reduced metrics
    for name in ("fname_per_frame.npy", "pose_per_frame.npy", "PVE_per_frame.npy"):
        a, b = np.load(one_dir / name), np.load(two_dir / name)
        assert a.shape == b.shape and a.shape[0] == 5
        if a.dtype.kind in "US":
            assert a.tolist() == b.tolist()
        else:
            assert np.abs(a - b).max() <= 1e-5 * max(1.0, np.abs(a).max())
