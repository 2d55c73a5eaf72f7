#!/usr/bin/env python
"""Benchmark of the per-image inference hot path on MI355X (contract: see the round prompt / DESIGN.md).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A step = one pass of the hot path over one batch of synthetic proxy representations per GPU:
ResNet-18 encoder -> distribution head -> matrix-Fisher sampling (num_samples=100) -> SMPL forward over
B*(N+2) meshes -> per-vertex uncertainty (BASELINE.json configs[1]: batch 64, 256x256, num_samples 100).
Inputs are resident in HBM before the timed region.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from hierarchicalprobabilistic3dhuman_amd import configs, smpl_data, sharding, sampling_utils  # noqa: E402
from hierarchicalprobabilistic3dhuman_amd.poseMF_shapeGaussian_net import PoseMFShapeGaussianNet  # noqa: E402
from hierarchicalprobabilistic3dhuman_amd.predict_poseMF_shapeGaussian_net import infer, InferencePipeline  # noqa: E402
from hierarchicalprobabilistic3dhuman_amd.smpl_official import SMPL  # noqa: E402

LBS_BYTES_PER_MESH = 166896          # 82,680 (v_posed) + 1,536 (A) + 82,680 (verts)
BLEND_FLOP_PER_MESH = 2 * (207 + 10) * 3 * 6890   # pose blend 8.56 MFLOP + shape blend 0.41 MFLOP per mesh
HBM_PEAK_GBS = 8000.0                # MI355X_MICROARCH.md: HBM3E 8 TB/s
ENCODER_GFLOP_PER_IMAGE = 6.279      # 3.139 GMAC per 18x256x256 image
MFMA_FP32_PEAK_TF = 157.3            # MI355X_MICROARCH.md: dense fp32 matrix peak
INPUT_SETS = 2                       # different input batches rotated through the timed loop
SET_STRIDE = 1 << 16                 # global image index distance between the input sets (> any global batch)


def synthetic_inputs(lo, hi):
    """Proxy representations of global images lo..hi-1; each image has its own seed so the data of image i
    does not depend on how the batch is sharded."""
    xs = [torch.rand(18, 256, 256, generator=torch.Generator().manual_seed(1000 + i)) for i in range(lo, hi)]
    return torch.stack(xs)


def cpu_baseline(net_state, parents, num_samples, n_images):
    """The CPU oracle (a port of the reference's PyTorch-CPU path, validated against the imported reference)
    timed on this host's cores on a bounded sample of the same workload."""
    from oracle import ref_cpu as O
    model = smpl_data.synthetic_smpl_model(0)
    params = O.SMPLParams(model, smpl_data.load_extra_joint_regressors(None), configs.SMPLX_EXTRA_VERTEX_IDS)
    x = synthetic_inputs(0, n_images)
    default_threads = torch.get_num_threads()
    # every hardware thread this process may use (affinity / cgroup quota: never oversubscribed).  Under torch.distributed.run the
    # default is ONE thread (OMP_NUM_THREADS=1); the other ranks are idle while rank 0 times the baseline, so it takes the host
    cores = max(1, min(os.cpu_count() or 1, sharding.effective_cpus()))
    torch.set_num_threads(cores)
    with torch.no_grad():
        torch.manual_seed(0)
        O.infer(net_state, params, parents, x[:1], num_samples)          # warm (thread pools, allocator)
        stages = {}
        t0 = time.perf_counter()
        O.infer(net_state, params, parents, x, num_samples, timings=stages)
        dt = time.perf_counter() - t0
        # the same port on ONE host thread, on a smaller sample
        n1 = min(2, n_images)
        torch.set_num_threads(1)
        try:
            t1 = time.perf_counter()
            O.infer(net_state, params, parents, x[:n1], num_samples)
            dt1 = time.perf_counter() - t1
        finally:
            torch.set_num_threads(default_threads)
        # the reference's OWN operating point: one image per call, num_samples = 50, image after image
        # (predict/predict_poseMF_shapeGaussian_net.py:58-59 loop, :103-165 per image) -- what secondary.latency_b1 is the GPU side of
        lat = {}
        for label, threads, reps in (("all_threads", cores, 6), ("single_thread", 1, 3)):
            torch.set_num_threads(threads)
            try:
                O.infer(net_state, params, parents, x[:1], 50)
                ts = []
                for i in range(reps):
                    t1 = time.perf_counter()
                    O.infer(net_state, params, parents, x[i % n_images:i % n_images + 1], 50)
                    ts.append(time.perf_counter() - t1)
            finally:
                torch.set_num_threads(default_threads)
            ts.sort()
            lat[label] = {"median_ms": ts[len(ts) // 2] * 1e3, "min_ms": ts[0] * 1e3, "images_per_s": 1.0 / ts[len(ts) // 2], "cores": threads, "reps": reps}
        lat["note"] = ("oracle/ref_cpu.infer on ONE image per call, num_samples=50, calls looped one after another like "
                       "predict/predict_poseMF_shapeGaussian_net.py:58-59,103-165 (the CPU side of secondary.latency_b1)")
    return {"value": n_images / dt, "unit": "images/s", "cores": cores, "kind": "port", "latency_b1": lat,
            "sample": "%d images (one batch of the metric's workload), num_samples=%d, oracle/ref_cpu.infer timed once after a 1-image "
                      "warm-up (%.2f s).  The port batches what the reference loops over: it makes ONE flattened SMPL call over all "
                      "B*(N+2) meshes and one batched encoder / head pass, where predict_poseMF_shapeGaussian_net.py handles one image "
                      "at a time -- so this baseline is FASTER than the reference's own CPU path would be" % (n_images, num_samples, dt),
            "host": "%d usable hardware threads (affinity / cgroup quota) of %d reported" % (sharding.effective_cpus(), os.cpu_count() or 0),
            "stage_seconds": {k: round(v, 4) for k, v in stages.items()},
            "single_thread": {"value": n1 / dt1, "unit": "images/s", "cores": 1,
                              "sample": "%d images, num_samples=%d (%.2f s)" % (n1, num_samples, dt1)}}


def live_traffic(kernel_regex, n_steps=3, extra=()):
    """HBM-side bytes per launch of the mesh kernel, MEASURED IN THIS RUN: two short rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE: they do
    not fit one pass) over a three-step sub-run of this very script, on this GPU, after the timed region; corrected as
    MI355X_MICROARCH.md prescribes for gfx950 (values in KB; fabric reads = 2 x FETCH_SIZE).  None when rocprofv3 is not on the PATH or a
    pass fails (the caller then falls back to the committed summary and says so)."""
    import csv
    import shutil
    import subprocess
    import tempfile
    if shutil.which("rocprofv3") is None:
        return None
    out = tempfile.mkdtemp(prefix="hps_traffic_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    kb = {}
    try:
        for c in ("FETCH_SIZE", "WRITE_SIZE"):
            cmd = ["rocprofv3", "--pmc", c, "--kernel-trace", "--kernel-include-regex", kernel_regex, "--output-format", "csv", "-d", out,
                   "-o", "pass_" + c, "--", sys.executable, os.path.abspath(__file__), "--steps", str(n_steps), "--warmup", "1",
                   "--cpu-images", "0", "--from-rgb-steps", "0", "--latency-reps", "0", "--stress-steps", "0", "--lbs-unfused-reps", "0",
                   "--live-traffic", "off", "--split-steps", "0"] + list(extra)
            p = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=180)
            path = None
            for root_, _, files in os.walk(out):
                for f in files:
                    if f.endswith("pass_%s_counter_collection.csv" % c):
                        path = os.path.join(root_, f)
            if path is None:
                return None
            vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(path)) if r["Counter_Name"] == c]
            vals = vals[1:] if len(vals) > 1 else vals            # the first launch of the sub-run is its warm-up step
            if not vals:
                return None
            kb[c] = (sum(vals) / len(vals), len(vals))
    except (OSError, subprocess.SubprocessError, ValueError, KeyError):
        return None
    finally:
        shutil.rmtree(out, ignore_errors=True)
    return {"hbm_bytes_per_launch": (2.0 * kb["FETCH_SIZE"][0] + kb["WRITE_SIZE"][0]) * 1024.0,
            "FETCH_SIZE_KB_per_launch": kb["FETCH_SIZE"][0], "WRITE_SIZE_KB_per_launch": kb["WRITE_SIZE"][0],
            "launches": min(kb["FETCH_SIZE"][1], kb["WRITE_SIZE"][1]),
            "method": "rocprofv3 --pmc {FETCH_SIZE | WRITE_SIZE} --kernel-trace --kernel-include-regex %s, two passes of a %d-step sub-run of "
                      "bench.py on this GPU after the timed region; bytes = (2 x FETCH_SIZE + WRITE_SIZE) KB (the gfx950 correction of "
                      "MI355X_MICROARCH.md)" % (kernel_regex, n_steps)}


def fused_path_ok(smpl):
    """The shared-shape fused mesh kernel is what infer(use_mean_shape=True) runs (the bf16x3 arithmetic exists for that form only)."""
    return bool(smpl.fused_mesh and smpl.shared_shape and smpl.picked_joints)


def workload_name(B, N, world):
    """Which BASELINE.json configuration the arguments select (the label is derived, never hard-coded)."""
    if B == 64 and N == 100:
        return "BASELINE configs[1]" if world == 1 else ("BASELINE configs[2]" if world == 8 else
                                                         "BASELINE configs[1] per GPU, %d GPUs (configs[2] is the 8-GPU case)" % world)
    if N == 1000:
        return "BASELINE configs[4] (num_samples=1000 stress)" + ("" if B == 16 else ", batch %d instead of 16" % B)
    return "custom (not a BASELINE configuration)"


def self_launch(args):
    """`python bench.py --gpus N` with N > 1 and no RANK / WORLD_SIZE in the environment: start the N ranks ourselves -- the
    command the driver would otherwise have to wrap (`python -m torch.distributed.run --nnodes=1 --nproc-per-node N ...`), on a
    free port of 127.0.0.1 -- relay rank 0's single JSON line on stdout (everything else the children print goes to stderr) and
    return their exit status."""
    import socket
    import subprocess
    have = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if args.backend == "nccl" and have < args.gpus:
        sys.stderr.write("bench.py --gpus %d: only %d HIP device(s) visible and RCCL (--backend nccl) needs one device per rank; run on a "
                         "node with %d GPUs, or pass --backend gloo for a functional run whose ranks share devices\n"
                         % (args.gpus, have, args.gpus))
        return 2
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: RCCL between processes needs it on this driver
    env.setdefault("OMP_NUM_THREADS", "1")                 # what torchrun would set anyway; the ranks' host side is one thread
    p = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE)
    lines = 0
    for raw in p.stdout:
        text = raw.decode(errors="replace")
        if text.startswith("{"):
            sys.stdout.write(text)
            sys.stdout.flush()
            lines += 1
        else:
            sys.stderr.write(text)
    rc = p.wait()
    if rc == 0 and lines != 1:
        sys.stderr.write("bench.py: expected ONE JSON line from rank 0, saw %d\n" % lines)
        return 3
    return rc


def _capi_mesh_cus(pipe):
    """CUs per XCD of the mesh partition of a pipeline that runs encoder and mesh kernels on disjoint CU subsets."""
    return getattr(pipe, "_mesh_cus", 32)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--batch", type=int, default=64, help="images per GPU per step")
    ap.add_argument("--num-samples", type=int, default=100)
    ap.add_argument("--no-pipeline", action="store_true", help="run the steps strictly one after another on one stream")
    ap.add_argument("--early-relayout", action="store_true", help="A/B: enqueue the input relayout beside the previous batch's mesh kernels")
    ap.add_argument("--mesh-overlap", choices=("auto", "on", "off"), default="auto",
                    help="let the mesh kernel run beside the neighbouring batches' encoders (auto: for batches below 32 images, whose "
                         "encoder cannot fill the chip)")
    ap.add_argument("--encoder-cus", type=int, default=None, help="CUs per XCD of the encoder's partition when the mesh kernel overlaps (0 = shared CUs)")
    ap.add_argument("--event-every", type=int, default=1,
                    help="record the HIP timing events around the mesh kernel / encoder on every n-th step only (0 = never: no roofline from this run)")
    ap.add_argument("--no-inline-mesh", action="store_true", help="A/B: the mesh kernel on the caller's stream with an event on either side (round 4) instead of on the encoder's stream")
    ap.add_argument("--stem-route", choices=("default", "nchw", "frames"), default="default",
                    help="A/B: 'nchw' = the stem gathers its windows from the NCHW input itself (no phase split), 'frames' = hps_stem_phase_split + "
                         "the frame-fed stem (round 5); default: ResNet.stem_reads_nchw as the model sets it")
    ap.add_argument("--side-on-caller-stream", action="store_true",
                    help="A/B: joint regression + uncertainty pass on the caller's stream, racing the next encoder's first kernel (round 5), "
                         "instead of behind the mesh kernel on the encoder's stream")
    ap.add_argument("--separate-downsample", action="store_true", help="A/B: the 1x1/2 down-sample of layer2-4's first block as a launch of its own (round 5) instead of inside the block's 3x3/2 convolution's launch")
    ap.add_argument("--unfused-pool", action="store_true", help="A/B: stem and max pool as two kernels (round 4) instead of the pool in the stem kernel's epilogue")
    ap.add_argument("--head-cus", type=int, default=None,
                    help="exclusive schedule: CUs per XCD reserved for the head's kernels (the mesh kernel runs on the others); 0 = shared CUs")
    ap.add_argument("--gather-joints", action="store_true", help="A/B: the joint regression gathers its vertices from the meshes (round 4) instead of reading the mesh kernel's compact side output")
    ap.add_argument("--head-high-priority", action="store_true", help="A/B: the head's stream at high priority (rounds 3-5) instead of normal priority")
    ap.add_argument("--separate-joints", action="store_true", help="A/B: joint regression and uncertainty pass as two launches (round 5) instead of one (hps_joints_and_uncertainty)")
    ap.add_argument("--per-mesh-shape-blend", action="store_true", help="A/B: the K = 217 form of the fused mesh kernel (shape blend inside the GEMM, once per mesh: round 5) instead of the shared-shape form (K = 207, shape blend once per image)")
    ap.add_argument("--mesh-arith", choices=("f32", "bf16x3"), default="f32",
                    help="arithmetic of the shared-shape mesh kernel's pose blend GEMM in the HEADLINE run: f32 = v_mfma_f32_32x32x2_f32 (default, the "
                         "reference's arithmetic type in every instruction); bf16x3 = fp32 operands as three bf16 pieces, six exact piece products on the "
                         "bf16 matrix pipe, fp32 accumulation (fp32 accuracy, other bits; SMPL.mesh_arith)")
    ap.add_argument("--split-steps", type=int, default=16,
                    help="after the timed region: pipelined steps with SMPL.mesh_arith = 'bf16x3' for secondary.mesh_bf16x3 (0 = skip)")
    ap.add_argument("--unfused-mesh", action="store_true", help="blend GEMM + LBS as two kernels (the unfused definition) instead of the fused mesh kernel")
    ap.add_argument("--trace-steps", action="store_true", help="print host-side per-step times to stderr (debugging)")
    ap.add_argument("--cpu-images", type=int, default=64, help="images in the CPU-baseline sample (0 = skip); 64 = one full batch, SURVEY 8(d)")
    ap.add_argument("--backend", choices=("nccl", "gloo"), default="nccl",
                    help="process-group backend when launched by torch.distributed.run: nccl (= RCCL over xGMI, the product "
                         "setting) or gloo (functional runs with more ranks than GPUs: ranks share devices, collectives are "
                         "host-staged)")
    ap.add_argument("--as-rank", type=int, nargs=2, metavar=("R", "W"), default=None,
                    help="single process, no process group: do exactly the work rank R of a W-rank job would do (its image "
                         "shard, its Philox offsets) -- what the multi-rank tests compare the per-rank checksums against")
    ap.add_argument("--from-rgb-steps", type=int, default=16,
                    help="after the timed region: steps of the PCIe-inclusive path from host RGB crops + keypoints for secondary.from_rgb (0 = skip)")
    ap.add_argument("--from-rgb-variant", choices=("default", "no-copy", "nchw"), default="default",
                    help="diagnostic: 'no-copy' runs the from-RGB loop on device-resident crops (isolates the cost of the H2D copies); "
                         "'nchw' builds the (B,18,D,D) tensor and lets the encoder split it into phase frames (the round-4 route)")
    ap.add_argument("--latency-reps", type=int, default=40,
                    help="after the timed region: batch-1, num_samples=50 calls timed one by one for secondary.latency_b1 (0 = skip)")
    ap.add_argument("--stress-steps", type=int, default=8,
                    help="after the timed region (one GPU only): pipelined steps of BASELINE configs[4] (batch 16, num_samples 1000) "
                         "for secondary.stress_n1000 (0 = skip)")
    ap.add_argument("--live-traffic", choices=("auto", "off"), default="auto",
                    help="auto (one GPU only): measure roofline.traffic in this run with two rocprofv3 --pmc passes over a three-step sub-run "
                         "(~15 s); off / unavailable: read it from the committed PMC summary under profiles/ and flag it traffic_imported")
    ap.add_argument("--lbs-unfused-reps", type=int, default=12,
                    help="after the timed region: launches of the unfused blend + LBS pair timed for secondary.lbs_unfused (0 = skip)")
    args = ap.parse_args()

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (HIP) device; there is no CPU fallback for the product path")
    if args.gpus > 1 and args.as_rank is None and not ("RANK" in os.environ and "WORLD_SIZE" in os.environ):
        raise SystemExit(self_launch(args))                 # bare `python bench.py --gpus N`: one process per GPU, started here
    rank, world, local_rank = sharding.init_distributed(args.backend)
    if world != args.gpus:
        raise SystemExit("bench.py --gpus %d must be launched with %d ranks (torch.distributed.run), got WORLD_SIZE=%d"
                         % (args.gpus, args.gpus, world))
    dev_index = sharding.local_device(local_rank)       # local_rank % visible devices (gloo runs may share a device)
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    B, N = args.batch, args.num_samples
    shard_rank, shard_world = (rank, world) if args.as_rank is None else tuple(args.as_rank)

    cfg = configs.get_cfg_defaults()
    torch.manual_seed(0)
    net = PoseMFShapeGaussianNet(configs.SMPL_PARENTS, cfg).eval()     # default init under seed 0 (random weights)
    net_state = {k: v.clone() for k, v in net.state_dict().items()} if rank == 0 else None
    net = net.to(dev)
    smpl = SMPL(smpl_data.synthetic_smpl_model(0), batch_size=1, gender="neutral", num_betas=10).to(dev)
    smpl.fused_mesh = not args.unfused_mesh
    smpl.picked_joints = not args.gather_joints
    smpl.shared_shape = not args.per_mesh_shape_blend
    smpl.mesh_arith = args.mesh_arith
    if args.separate_joints:
        from hierarchicalprobabilistic3dhuman_amd import predict_poseMF_shapeGaussian_net as _pm
        _pm.FUSE_JOINTS_AND_UNCERTAINTY = False

    lo, hi = sharding.shard_range(B * shard_world, shard_rank, shard_world)   # weak scaling: B images per GPU
    # INPUT_SETS different global batches rotate through the steps (step i reads set i % INPUT_SETS: no step re-reads the
    # tensor the step before it read); set k holds the global images k * SET_STRIDE + [lo, hi) -- still a function of the global
    # image index only, so a rank's data does not depend on the sharding
    set_stride = SET_STRIDE
    xs = [synthetic_inputs(lo + k * set_stride, hi + k * set_stride).to(dev) for k in range(INPUT_SETS)]
    x = xs[0]
    offset_of = lambda step: lo + (step % INPUT_SETS) * set_stride          # Philox key: global index of the batch's first image

    # Steps are software-pipelined over three HIP streams: the encoder of step i+1 is enqueued before the latency-bound
    # head of step i runs (InferencePipeline).  Every step does the full work; exactly args.steps batches are
    # submitted and finished inside the timed region.
    pipe = InferencePipeline(net, smpl, num_samples=N, use_mean_shape=True)
    pipe.early_relayout = args.early_relayout
    pipe.exclusive_mesh = {"auto": None, "on": False, "off": True}[args.mesh_overlap]
    if args.encoder_cus is not None:
        pipe.encoder_cus = args.encoder_cus
    if args.head_cus is not None:
        pipe.head_cus = args.head_cus
    pipe.inline_mesh = not args.no_inline_mesh
    net.image_encoder.fused_pool = not args.unfused_pool
    net.image_encoder.fold_downsample = not args.separate_downsample
    pipe.inline_side = not args.side_on_caller_stream
    if args.head_high_priority:
        pipe.head_stream = torch.cuda.Stream(priority=-1)
    if args.stem_route != "default":
        net.image_encoder.stem_reads_nchw = args.stem_route == "nchw"

    step_marks = []

    ev_lists = {"lbs": None, "enc": None}

    def sample_events(step):
        # timing events on every args.event_every-th step only: an event record is a marker packet the command processor handles in
        # ~15 us, and with the mesh kernel on the encoder's stream the markers sit on the critical path
        on = args.event_every > 0 and step % args.event_every == 0
        smpl.lbs_events = ev_lists["lbs"] if on else None
        return on

    def run_steps(first, count, on_result=None):
        # input_ready=False: the inputs are resident and complete before the loop starts (nothing produces them on a stream)
        pipe.enc_events = ev_lists["enc"] if sample_events(first) else None
        ticket = pipe.submit(xs[first % INPUT_SETS], input_ready=False)
        res = None
        for i in range(count):
            pipe.enc_events = ev_lists["enc"] if (args.event_every > 0 and (first + i + 1) % args.event_every == 0) else None
            nxt = pipe.submit(xs[(first + i + 1) % INPUT_SETS], input_ready=False) if i + 1 < count else None
            sample_events(first + i)
            res = pipe.finish(ticket, seed=1234 + first + i, image_offset=offset_of(first + i), after=nxt)
            if on_result is not None:
                on_result(res)
            ticket = nxt
            if args.trace_steps:
                step_marks.append(time.perf_counter())
        return res

    distributed = torch.distributed.is_available() and torch.distributed.is_initialized()
    barrier = sharding.barrier

    # the loop runs on the stream the pipeline asks for (small batches: the mesh partition's stream, see InferencePipeline)
    loop_stream = pipe.caller_stream(B) if not args.no_pipeline else torch.cuda.current_stream()
    torch.cuda.current_stream().synchronize()
    torch.cuda.set_stream(loop_stream)
    # warm-up runs everything the timed region runs (including the metric accumulation and the collective), so
    # no kernel is loaded for the first time inside the timed region
    warm_sums = torch.zeros(4, dtype=torch.float64, device=dev)
    if not args.no_pipeline:
        run_steps(0, args.warmup, lambda r: sharding.batch_metric_sums(r, accumulate=warm_sums))
    else:
        for i in range(args.warmup):
            sharding.batch_metric_sums(infer(net, smpl, xs[i % INPUT_SETS], num_samples=N, use_mean_shape=True, seed=1234 + i,
                                             image_offset=offset_of(i)), accumulate=warm_sums)
    sharding.gather_metric_sums(warm_sums)
    torch.cuda.synchronize()
    smpl.lbs_events = ev_lists["lbs"] = []
    pipe.enc_events = ev_lists["enc"] = []
    if args.trace_steps:
        pipe.trace = []
    sampling_utils.launch_events = []
    sums = torch.zeros(4, dtype=torch.float64, device=dev)
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    def accumulate(res):
        sharding.batch_metric_sums(res, accumulate=sums)      # the running total is formed by the checksum's own second launch

    if not args.no_pipeline:
        run_steps(args.warmup, args.steps, accumulate)
    else:
        for i in range(args.steps):
            accumulate(infer(net, smpl, xs[(args.warmup + i) % INPUT_SETS], num_samples=N, use_mean_shape=True,
                             seed=1234 + args.warmup + i, image_offset=offset_of(args.warmup + i)))
    per_rank, total = sharding.gather_metric_sums(sums)                 # the one collective of the run
    torch.cuda.synchronize()
    barrier()
    dt = sharding.all_reduce_max(time.perf_counter() - t0)               # the slowest rank's time

    # LBS kernel time from HIP events recorded around each launch on the launch stream
    M = B * (N + 2)
    lbs_ms = [e0.elapsed_time(e1) for (m, e0, e1) in (ev_lists["lbs"] or []) if m == M]
    smpl.lbs_events = None
    lbs_avg_ms = sum(lbs_ms) / max(1, len(lbs_ms))
    achieved = LBS_BYTES_PER_MESH * M / (lbs_avg_ms * 1e-3) / 1e9 if lbs_ms else None

    def spread(ms):
        ms = sorted(ms)
        return {"median_ms": ms[len(ms) // 2] if len(ms) % 2 else 0.5 * (ms[len(ms) // 2 - 1] + ms[len(ms) // 2]),
                "min_ms": ms[0], "max_ms": ms[-1], "launches": len(ms)} if ms else None

    # secondary figures of SURVEY section 8(d), same method (HIP events on the launch stream, timed region only); the
    # encoder shares the GPU with the previous batch's head and uncertainty kernels while it runs
    enc_events_kept = list(ev_lists["enc"] or [])
    enc_ms = [e0.elapsed_time(e1) for (e0, e1) in enc_events_kept]
    smp_ms = [e0.elapsed_time(e1) for (e0, e1) in (sampling_utils.launch_events or [])]
    pipe.enc_events, sampling_utils.launch_events = None, None
    # The opt-in arithmetic of the mesh kernel under the same clock: the same pipelined steps with SMPL.mesh_arith = "bf16x3" (the pose
    # blend GEMM on the bf16 matrix pipe at fp32 accuracy: csrc/mesh_split.hip), a few steps RIGHT AFTER the timed region -- before the
    # other legs create their own streams and pipelines (behind the configs[4] leg's CU-partition queues the same steps measured 0-30 %
    # slower, run to run) --, wall clock + the kernel's own HIP events; and how far its vertices are from the fp32-MFMA kernel's on one
    # batch (same seed, same samples).
    mesh_bf16x3 = None
    if args.split_steps > 0 and world == 1 and not args.no_pipeline and args.mesh_arith == "f32" and fused_path_ok(smpl):
        ref = infer(net, smpl, x, num_samples=N, use_mean_shape=True, seed=31337, image_offset=lo)
        smpl.mesh_arith = "bf16x3"
        try:
            alt = infer(net, smpl, x, num_samples=N, use_mean_shape=True, seed=31337, image_offset=lo)
            dv = float((alt["verts_samples"] - ref["verts_samples"]).abs().max()) if "verts_samples" in alt else None
            dm = float((alt["verts_mode"] - ref["verts_mode"]).abs().max())
            du = float((alt["unc"] - ref["unc"]).abs().max())
            del ref, alt
            run_steps(0, 4)
            torch.cuda.synchronize()
            smpl.lbs_events = ev_lists["lbs"] = []
            pipe.enc_events = ev_lists["enc"] = []
            t_a = time.perf_counter()
            run_steps(4, args.split_steps)
            torch.cuda.synchronize()
            dt_b = time.perf_counter() - t_a
            b_mesh = [e0.elapsed_time(e1) for (m, e0, e1) in ev_lists["lbs"] if m == M]
            b_enc = [e0.elapsed_time(e1) for (e0, e1) in ev_lists["enc"]]
        finally:
            smpl.mesh_arith = "f32"
            smpl.lbs_events = pipe.enc_events = None
            ev_lists["lbs"] = ev_lists["enc"] = None
        bm = spread(b_mesh)
        mesh_bf16x3 = {"images_per_s": B * args.split_steps / dt_b, "ms_per_step": dt_b / args.split_steps * 1e3, "steps": args.split_steps,
                       "vs_headline": (B * args.split_steps / dt_b) / (B * world * args.steps / dt),
                       "mesh_kernel": "hps::mesh_split_kernel<4,24,4,0>", "mesh_kernel_ms": bm, "encoder_avg_ms": sum(b_enc) / max(1, len(b_enc)),
                       "max_abs_diff_vs_f32_m": {"verts_samples": dv, "verts_mode": dm, "vertex_uncertainty": du},
                       "blend_tflops_fp32_equivalent": (2 * 207 * 3 * 6890 * M / (bm["median_ms"] * 1e-3) / 1e12) if bm else None,
                       "issued_bf16_tflops": (6 * 2 * 208 * 3 * 6912 * M / (bm["median_ms"] * 1e-3) / 1e12) if bm else None,
                       "note": "opt-in (SMPL.mesh_arith = 'bf16x3', bench.py --mesh-arith bf16x3 makes it the headline's kernel): every fp32 operand of "
                               "the pose blend GEMM as three bf16 pieces (exact split), six exact piece products per product on "
                               "v_mfma_f32_32x32x16_bf16, fp32 accumulation -- as close to the float64 twin as the fp32-MFMA kernel "
                               "(tests/test_gpu_smpl.py::test_split_bf16_form_of_the_mesh_kernel), not the same bits.  The default keeps "
                               "the reference's arithmetic type in every instruction."}
    # SURVEY 8(d)'s LBS kernel itself (BASELINE's "LBS HBM GB/s"): the product path skins inside the blend GEMM's epilogue,
    # so after the timed region the same workload is run a few more steps with the unfused pair hps_smpl_blend + hps_smpl_lbs
    # (bit-identical vertices) and the hps_smpl_lbs launches are timed with HIP events on their stream, like the product kernel.
    lbs_unfused = None
    if args.lbs_unfused_reps > 0 and not args.unfused_mesh:
        smpl.fused_mesh, smpl.lbs_events = False, []
        for i in range(args.lbs_unfused_reps + 2):
            infer(net, smpl, x, num_samples=N, use_mean_shape=True, seed=4321 + i, image_offset=lo)
        torch.cuda.synchronize()
        ums = [e0.elapsed_time(e1) for (m, e0, e1) in smpl.lbs_events if m == M][2:]      # first two: warm-up
        smpl.fused_mesh, smpl.lbs_events = True, None
        if ums:
            st = spread(ums)
            gbs = LBS_BYTES_PER_MESH * M / (st["median_ms"] * 1e-3) / 1e9
            lbs_unfused = dict(st, kernel="hps::lbs_kernel<4,8,1>", bound="hbm", achieved=gbs, peak=HBM_PEAK_GBS, unit="GB/s",
                               frac=gbs / HBM_PEAK_GBS, algorithmic_bytes_per_launch=LBS_BYTES_PER_MESH * M,
                               note="hps_smpl_lbs alone (v_posed read + A read + verts write: SURVEY 8(d)'s definition) on the "
                                    "bench workload, launched after the timed region; not part of the product path, which fuses "
                                    "skinning into the blend GEMM")
    # Latency at the reference's own operating point (run_predict.py -> predict/predict_poseMF_shapeGaussian_net.py:58-59,103-165:
    # ONE image at a time, num_samples = 50): input resident, infer() issued and waited for, wall clock per image.
    latency_b1 = None
    if args.latency_reps > 0 and rank == 0:
        x1, n1 = xs[0][:1].contiguous(), 50
        lat = []
        for i in range(args.latency_reps + 5):
            torch.cuda.synchronize()
            t_a = time.perf_counter()
            infer(net, smpl, x1, num_samples=n1, use_mean_shape=True, seed=99 + i, image_offset=lo)
            torch.cuda.synchronize()
            lat.append((time.perf_counter() - t_a) * 1e3)
        st = spread(lat[5:])
        # the same calls with the encoder in latency mode (ResNet.set_latency_mode: direct kernels, many K slices -- a per-model
        # switch for one-image-at-a-time deployments; ~1.5x slower than the default at batch 64, hence not the bench's mode)
        lat2 = []
        net.set_latency_mode(True)
        try:
            for i in range(args.latency_reps + 5):
                torch.cuda.synchronize()
                t_a = time.perf_counter()
                infer(net, smpl, x1, num_samples=n1, use_mean_shape=True, seed=99 + i, image_offset=lo)
                torch.cuda.synchronize()
                lat2.append((time.perf_counter() - t_a) * 1e3)
        finally:
            net.set_latency_mode(False)
        st2 = spread(lat2[5:])
        # ... and the same calls replayed from a hipGraph (GraphedInfer: what predict_poseMF_shapeGaussian_net does for batch_size <= 2):
        # one graph launch per image instead of ~50 kernel launches; same kernels, same bits
        from hierarchicalprobabilistic3dhuman_amd.predict_poseMF_shapeGaussian_net import GraphedInfer
        graph_ms = {}
        for mode_name, on in (("latency", True), ("throughput", False)):
            net.set_latency_mode(on)
            try:
                g1 = GraphedInfer(net, smpl, batch=1, num_samples=n1, slots=1)
                tl = []
                for i in range(args.latency_reps + 5):
                    torch.cuda.synchronize()
                    t_a = time.perf_counter()
                    g1(x1, seed=99 + i, image_offset=lo)
                    torch.cuda.synchronize()
                    tl.append((time.perf_counter() - t_a) * 1e3)
                graph_ms[mode_name] = spread(tl[5:])
                del g1
            finally:
                net.set_latency_mode(False)
        latency_b1 = {"median_ms": st2["median_ms"], "min_ms": st2["min_ms"], "max_ms": st2["max_ms"], "reps": st2["launches"],
                      "images_per_s": 1e3 / st2["median_ms"], "batch": 1, "num_samples": n1,
                      "model_mode": "latency (PoseMFShapeGaussianNet.set_latency_mode(True): direct kernels with many K slices, wide head workgroups)",
                      "throughput_mode_median_ms": st["median_ms"],
                      "graph_median_ms": graph_ms["latency"]["median_ms"], "graph_min_ms": graph_ms["latency"]["min_ms"],
                      "throughput_mode_graph_median_ms": graph_ms["throughput"]["median_ms"],
                      "note": "one image per call, host wall clock from issue to completion (torch.cuda.synchronize), input resident "
                              "in HBM; the reference's run_predict operating point.  median_ms: encoder in latency mode (what a "
                              "batch-1 deployment selects), launches issued one by one; throughput_mode_median_ms: the default kernels "
                              "of the headline; graph_*: the same call replayed from a hipGraph (GraphedInfer, one slot: includes the "
                              "input's copy into the graph's static buffer)"}
    # The path from where the REFERENCE starts (predict/predict_poseMF_shapeGaussian_net.py:61-100): RGB crops (3 x 256 x 256) and
    # 17 keypoints per image in page-locked HOST memory -> staged non-blocking H2D copy on a copy stream (786 KB per image
    # instead of the 4.7 MB of a finished proxy representation) -> Canny edge map + Gaussian heat-maps on the device
    # (hps_canny_edge_map, hps_proxy_rep) -> the same pipelined step.  PCIe-inclusive; reported beside the headline, never as it.
    from_rgb = None
    if args.from_rgb_steps > 0 and not args.no_pipeline:
        from hierarchicalprobabilistic3dhuman_amd.canny_edge_detector import CannyEdgeDetector
        from hierarchicalprobabilistic3dhuman_amd.predict_poseMF_shapeGaussian_net import StagedUpload, proxy_representation
        canny = CannyEdgeDetector(cfg.DATA.EDGE_NMS, cfg.DATA.EDGE_GAUSSIAN_STD, cfg.DATA.EDGE_GAUSSIAN_SIZE, cfg.DATA.EDGE_THRESHOLD).to(dev)
        D = cfg.DATA.PROXY_REP_SIZE
        host_sets = []
        for k in range(INPUT_SETS):
            g = torch.Generator().manual_seed(77 + k + 1000 * shard_rank)
            rgb = torch.nn.functional.interpolate(torch.rand(B, 3, D // 8, D // 8, generator=g), size=(D, D), mode="bilinear",
                                                  align_corners=False) + 0.05 * torch.rand(B, 3, D, D, generator=g)
            j2d = torch.rand(B, 17, 2, generator=g) * D
            vis = (torch.rand(B, 17, generator=g) > 0.15).float()
            host_sets.append([t.contiguous().pin_memory() for t in (rgb, j2d, vis)])
        stager = StagedUpload(slots=2)
        # the front end writes the proxy representation straight into the Winograd stem's phase frames (hps_proxy_rep_phase_frames):
        # no (B,18,D,D) tensor, no hps_stem_phase_split on this path ("nchw": the round-4 route through the NCHW tensor, for A/B)
        enc_direct = net.image_encoder if args.from_rgb_variant != "nchw" else None

        resident = [[t.to(dev) for t in hs] for hs in host_sets] if args.from_rgb_variant == "no-copy" else None

        def rgb_step(k):
            if resident is not None:          # diagnostic: the same loop without the H2D copies
                rgb_d, j_d, v_d = resident[k % INPUT_SETS]
                return pipe.submit(make_input=lambda: proxy_representation(rgb_d, j_d, v_d, canny, cfg, encoder=enc_direct), input_ready=False)
            (rgb_d, j_d, v_d), ready = stager.upload(host_sets[k % INPUT_SETS])
            t = pipe.submit(make_input=lambda: proxy_representation(rgb_d, j_d, v_d, canny, cfg, encoder=enc_direct), input_ready=ready)
            stager.release(t[1])
            return t

        def rgb_steps(first, count, sink):
            ticket = rgb_step(first)
            for i in range(count):
                nxt = rgb_step(first + i + 1) if i + 1 < count else None
                sharding.batch_metric_sums(pipe.finish(ticket, seed=777 + first + i, image_offset=lo, after=nxt), accumulate=sink)
                ticket = nxt

        # warm-up = one whole untimed leg (at least 8 steps): both upload slots, both input sets, the page-locked buffers' first
        # DMA mappings and the front-end kernels' code objects have all been used before the first timed leg (round 4 warmed 4
        # steps and its first leg read 12.6 k images/s beside 18.9 / 20.1 k in the driver's run)
        rgb_warm = max(8, args.from_rgb_steps)
        rgb_steps(0, rgb_warm, torch.zeros(4, dtype=torch.float64, device=dev))
        torch.cuda.synchronize()
        barrier()
        # three legs of --from-rgb-steps steps each; the median leg is reported and all three are listed
        legs = []
        for leg in range(3):
            pipe.enc_events, smpl.lbs_events = [], []
            leg_sums = torch.zeros(4, dtype=torch.float64, device=dev)
            torch.cuda.synchronize()
            barrier()
            t_a = time.perf_counter()
            rgb_steps(rgb_warm + leg * args.from_rgb_steps, args.from_rgb_steps, leg_sums)
            torch.cuda.synchronize()
            barrier()
            dt_leg = sharding.all_reduce_max(time.perf_counter() - t_a)
            legs.append((dt_leg, [e0.elapsed_time(e1) for (e0, e1) in pipe.enc_events],
                         [e0.elapsed_time(e1) for (m, e0, e1) in smpl.lbs_events if m == M], leg_sums))
        pipe.enc_events, smpl.lbs_events = None, None
        leg_rates = [B * world * args.from_rgb_steps / l[0] for l in legs]
        dt_rgb, rgb_enc_ms, rgb_mesh_ms, rgb_sums = sorted(legs, key=lambda l: l[0])[1]
        h2d = sum(t.numel() * t.element_size() for t in host_sets[0])
        from_rgb = {"images_per_s": B * world * args.from_rgb_steps / dt_rgb, "ms_per_step": dt_rgb / args.from_rgb_steps * 1e3,
                    "steps": args.from_rgb_steps, "legs_images_per_s": leg_rates, "h2d_bytes_per_step_per_gpu": h2d,
                    "checksum_images": float(rgb_sums[0]), "checksum_sum_unc": float(rgb_sums[1]),
                    "encoder_avg_ms": sum(rgb_enc_ms) / max(1, len(rgb_enc_ms)),
                    "mesh_kernel_avg_ms": sum(rgb_mesh_ms) / max(1, len(rgb_mesh_ms)),
                    "input_route": "phase frames written directly (hps_proxy_rep_phase_frames)" if enc_direct is not None else "NCHW tensor + hps_stem_phase_split",
                    "note": "PCIe-inclusive: page-locked host RGB crops + 17 keypoints + visibility -> non-blocking H2D on a copy "
                            "stream (two device slots) -> hps_canny_edge_map + hps_proxy_rep_phase_frames on the encoder's stream -> the same "
                            "pipelined step as the headline; median of three legs of %d steps (all listed) after %d warm-up steps, wall clock"
                            % (args.from_rgb_steps, rgb_warm),
                    "legs_spread": (max(leg_rates) - min(leg_rates)) / max(leg_rates)}
    # BASELINE configs[4] under the same clock: batch 16, num_samples 1000 = 16 032 meshes per step, the
    # pipelined loop with the schedule InferencePipeline picks for it (encoder and mesh kernels on disjoint CU subsets), a few
    # steps after the timed region.  Reported: images/s, the sampler's proposal rate, the mesh kernel against the blend GEMM's
    # MFMA roofline, the one-sweep uncertainty kernel and the unfused LBS kernel against the HBM roofline.
    stress = None
    if args.stress_steps > 0 and world == 1 and not args.no_pipeline:
        Bs, Ns = 16, 1000
        Ms = Bs * (Ns + 2)
        sx = [xs[k % INPUT_SETS][:Bs].contiguous() for k in range(2)]
        spipe = InferencePipeline(net, smpl, num_samples=Ns, use_mean_shape=True)
        s_stream = spipe.caller_stream(Bs)
        torch.cuda.synchronize()
        with torch.cuda.stream(s_stream):
            def stress_steps(first, count):
                ticket = spipe.submit(sx[first % 2], input_ready=False)
                for i in range(count):
                    nxt = spipe.submit(sx[(first + i + 1) % 2], input_ready=False) if i + 1 < count else None
                    r = spipe.finish(ticket, seed=555 + first + i, image_offset=lo, after=nxt)
                    ticket = nxt
                return r
            stress_steps(0, 3)
            torch.cuda.synchronize()
            spipe.enc_events, smpl.lbs_events, sampling_utils.launch_events, sampling_utils.unc_events = [], [], [], []
            t_a = time.perf_counter()
            r_last = stress_steps(3, args.stress_steps)
            torch.cuda.synchronize()
            dt_s = time.perf_counter() - t_a
            s_mesh = [e0.elapsed_time(e1) for (m, e0, e1) in smpl.lbs_events if m == Ms]
            s_enc = [e0.elapsed_time(e1) for (e0, e1) in spipe.enc_events]
            s_smp = [e0.elapsed_time(e1) for (e0, e1) in sampling_utils.launch_events]
            s_unc = [e0.elapsed_time(e1) for (b_, n_, e0, e1) in sampling_utils.unc_events if (b_, n_) == (Bs, Ns)]
            spipe.enc_events, smpl.lbs_events, sampling_utils.launch_events, sampling_utils.unc_events = None, None, None, None
            finite = bool(torch.isfinite(r_last["unc"]).all())
            # ... and the same steps with the opt-in bf16x3 arithmetic of the mesh kernel (this configuration is the one the mesh kernel dominates)
            s_split = None
            # the unfused LBS kernel at this size (SURVEY 8(d)'s definition), a few sequential calls on the same stream
            smpl.fused_mesh, smpl.lbs_events = False, []
            for i in range(5):
                infer(net, smpl, sx[0], num_samples=Ns, use_mean_shape=True, seed=4321 + i, image_offset=lo)
            torch.cuda.synchronize()
            s_lbs = [e0.elapsed_time(e1) for (m, e0, e1) in smpl.lbs_events if m == Ms][2:]
            smpl.fused_mesh, smpl.lbs_events = True, None
        # ... and the same steps with the opt-in bf16x3 arithmetic of the mesh kernel, on a pipeline of its own (the CU partition follows the
        # arithmetic: 12 of every XCD's 32 CUs for the encoder instead of 8, InferencePipeline.encoder_cus)
        if args.split_steps > 0 and args.mesh_arith == "f32" and fused_path_ok(smpl):
            smpl.mesh_arith = "bf16x3"
            try:
                bpipe = InferencePipeline(net, smpl, num_samples=Ns, use_mean_shape=True)
                b_stream = bpipe.caller_stream(Bs)
                torch.cuda.synchronize()
                with torch.cuda.stream(b_stream):
                    def bsteps(first, count):
                        ticket = bpipe.submit(sx[first % 2], input_ready=False)
                        for i in range(count):
                            nxt = bpipe.submit(sx[(first + i + 1) % 2], input_ready=False) if i + 1 < count else None
                            bpipe.finish(ticket, seed=555 + first + i, image_offset=lo, after=nxt)
                            ticket = nxt
                    bsteps(0, 3)
                    torch.cuda.synchronize()
                    smpl.lbs_events = []
                    t_b = time.perf_counter()
                    bsteps(3, args.stress_steps)
                    torch.cuda.synchronize()
                    dt_sb = time.perf_counter() - t_b
                    sb_mesh = [e0.elapsed_time(e1) for (m, e0, e1) in smpl.lbs_events if m == Ms]
                s_split = {"images_per_s": Bs * args.stress_steps / dt_sb, "ms_per_step": dt_sb / args.stress_steps * 1e3,
                           "mesh_kernel_median_ms": spread(sb_mesh)["median_ms"] if sb_mesh else None,
                           "schedule": ("encoder on %d of the 32 CUs of every XCD, mesh kernels on the other %d" % (32 - _capi_mesh_cus(bpipe), _capi_mesh_cus(bpipe))
                                        if bpipe.mesh_stream is not None else "shared CUs")}
                del bpipe
            finally:
                smpl.mesh_arith, smpl.lbs_events = "f32", None
        torch.cuda.set_stream(loop_stream)
        med = lambda v: spread(v)["median_ms"] if v else None
        mesh_ms, unc_ms, smp_s_ms, lbs_ms_s = med(s_mesh), med(s_unc), med(s_smp), med(s_lbs)
        unc_bytes = Bs * Ns * 6890 * 12 + Bs * 6890 * 4          # every sample vertex read once, the uncertainties written
        stress = {"workload": "BASELINE configs[4]: batch=%d, num_samples=%d (%d meshes per step), 1 GPU" % (Bs, Ns, Ms),
                  "images_per_s": Bs * args.stress_steps / dt_s, "ms_per_step": dt_s / args.stress_steps * 1e3, "steps": args.stress_steps,
                  "results_finite": finite,
                  "mesh_bf16x3": s_split,
                  "schedule": ("encoder on %d of the 32 CUs of every XCD, mesh kernels on the other %d" % (32 - _capi_mesh_cus(spipe), _capi_mesh_cus(spipe))
                               if spipe.mesh_stream is not None else "shared CUs"),
                  "encoder_avg_ms": sum(s_enc) / max(1, len(s_enc)),
                  "sampler": ({"median_ms": smp_s_ms, "proposals_per_s": Bs * 23 * 8 * Ns / (smp_s_ms * 1e-3),
                               "unit": "matrix-Fisher proposals/s as the reference counts them (8N per image and joint)"} if smp_s_ms else None),
                  "mesh_kernel": ({"median_ms": mesh_ms, "bound": "mfma", "achieved": BLEND_FLOP_PER_MESH * Ms / (mesh_ms * 1e-3) / 1e12,
                                   "peak": MFMA_FP32_PEAK_TF, "unit": "TFLOP/s", "frac": BLEND_FLOP_PER_MESH * Ms / (mesh_ms * 1e-3) / 1e12 / MFMA_FP32_PEAK_TF,
                                   "cus": 8 * _capi_mesh_cus(spipe) if spipe.mesh_stream is not None else 256,
                                   "note": "frac is against the WHOLE chip's fp32 MFMA peak; the kernel runs on `cus` of the 256 CUs in this schedule"}
                                  if mesh_ms else None),
                  "uncertainty_sweep1": ({"median_ms": unc_ms, "bound": "hbm", "achieved": unc_bytes / (unc_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS,
                                          "unit": "GB/s", "frac": unc_bytes / (unc_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "algorithmic_bytes_per_launch": unc_bytes,
                                          "note": "inside the pipelined loop (beside the next batch's encoder)"} if unc_ms else None),
                  "lbs_unfused": ({"median_ms": lbs_ms_s, "bound": "hbm", "achieved": LBS_BYTES_PER_MESH * Ms / (lbs_ms_s * 1e-3) / 1e9, "peak": HBM_PEAK_GBS,
                                   "unit": "GB/s", "frac": LBS_BYTES_PER_MESH * Ms / (lbs_ms_s * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                   "note": "hps_smpl_lbs alone at %d meshes (not on the product path)" % Ms} if lbs_ms_s else None)}
    secondary = {}
    if mesh_bf16x3:
        secondary["mesh_bf16x3"] = mesh_bf16x3
    if stress:
        secondary["stress_n1000"] = stress
    if from_rgb:
        secondary["from_rgb"] = from_rgb
    if latency_b1:
        secondary["latency_b1"] = latency_b1
    if enc_ms:
        t = sum(enc_ms) / len(enc_ms)
        tf = ENCODER_GFLOP_PER_IMAGE * B / t
        secondary["encoder"] = {"avg_ms": t, "tflops": tf, "peak": MFMA_FP32_PEAK_TF, "frac": tf / MFMA_FP32_PEAK_TF,
                                "unit": "TFLOP/s fp32 MFMA (6.279 GFLOP/image algorithmic, relayout/pools included in the time)"}
    if lbs_unfused:
        secondary["lbs_unfused"] = lbs_unfused
    if smp_ms:
        t = sum(smp_ms) / len(smp_ms)
        secondary["sampler"] = {"avg_ms": t, "proposals_per_s": B * 23 * 8 * N / (t * 1e-3),
                                "unit": "matrix-Fisher proposals/s (8N per image and joint, Philox)"}

    fused = bool(getattr(smpl, "fused_mesh", False))
    shared_shape = fused and bool(getattr(smpl, "shared_shape", False)) and smpl.picked_joints
    split_arith = shared_shape and smpl.mesh_arith == "bf16x3"
    mesh_kernel = "hps::mesh_split_kernel<4,24,4,0>" if split_arith else ("hps::mesh_fused_kernel<4,0,24,false,8,2,true,true>" if shared_shape else "hps::mesh_fused_kernel<4,0,24,false,5>") if fused else "hps::lbs_kernel<4,8,1>"
    k_required = 207 if shared_shape else 217            # K rows the launch has to multiply per mesh (the shared shape blend: once per IMAGE)
    # HBM traffic of the same kernel from the PMC counters: collected in separate rocprofv3 passes of this very
    # command (tools/collect_profiles.sh), corrected as MI355X_MICROARCH.md prescribes, committed under profiles/
    traffic, traffic_imported, traffic_detail = None, None, None
    pmc_name = "mesh_fused_pmc_latest.json" if fused else "lbs_pmc_latest.json"
    pmc_path = os.path.join(ROOT, "profiles", pmc_name)
    traffic_source = "profiles/%s (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes)" % pmc_name
    if args.live_traffic == "auto" and world == 1 and rank == 0 and args.as_rank is None and (B, N) == (64, 100) and not args.no_pipeline:
        torch.cuda.synchronize()
        traffic_detail = live_traffic(("mesh_split_kernel" if split_arith else "mesh_fused_kernel") if fused else "lbs_kernel",
                                      extra=["--mesh-arith", args.mesh_arith])
        if traffic_detail is not None:
            traffic, traffic_imported = traffic_detail["hbm_bytes_per_launch"], False
            traffic_source = traffic_detail["method"]
    if traffic is None and os.path.exists(pmc_path) and not split_arith:       # (the committed summary is the fp32 kernel's)
        with open(pmc_path) as f:
            pmc = json.load(f)
        if pmc.get("meshes_per_launch", M) == M:
            traffic = pmc.get("hbm_bytes_per_launch")
            traffic_imported = True

    if args.trace_steps and rank == 0 and pipe.trace and enc_ms:
        # device-side schedule of a few steady-state batches from HIP events (no profiler): batch j's encoder was the j-th
        # submit of the timed region, its head / mesh phases the j-th finish
        evs = enc_events_kept
        ref = evs[2][0]
        for j in range(2, min(8, len(pipe.trace))):
            t = pipe.trace[j]
            rel = lambda e: ref.elapsed_time(e)
            print("batch %2d: encoder %7.3f -> %7.3f | head %7.3f -> %7.3f | meshes %7.3f -> %7.3f  (ms)" % (
                j, rel(evs[j][0]), rel(evs[j][1]), rel(t["head0"]), rel(t["head1"]), rel(t["mesh0"]), rel(t["mesh1"])), file=sys.stderr)
    if args.trace_steps and rank == 0:
        marks = step_marks[-args.steps:]
        print("host ms between step completions:", ["%.2f" % ((b - a) * 1e3) for a, b in zip(marks[:-1], marks[1:])],
              file=sys.stderr)
    if rank == 0:
        images = B * world * args.steps
        out = {
            "metric": "images/sec end-to-end (num_samples=%d)" % N,
            "value": images / dt, "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "%s: batch=%d synthetic 18x256x256 proxy representations per GPU, "
                                   "num_samples=%d, neutral synthetic SMPL (6890 verts), random-init ResNet-18 + "
                                   "poseMF_shapeGaussian head (seed 0), Philox sampling" % (workload_name(B, N, world), B, N),
                       "input_sets": INPUT_SETS,
                       "images_per_gpu": B, "global_batch": B * world, "num_samples": N,
                       "meshes_per_step_per_gpu": M, "parallelism": "images sharded over %d GPU(s)" % world,
                       "mesh_kernel": "fused blend GEMM + LBS" if fused else "blend GEMM, then LBS",
                       "mesh_arith": smpl.mesh_arith if shared_shape else "f32",
                       "step_pipelining": "none" if args.no_pipeline else (
                           ("encoder of step i+1 (own stream) overlaps the latency-bound head of step i (high-priority stream); the mesh kernels (pose prep / blend + LBS / joints) of a batch run alone"
                            + ("; the head on %d CUs of every XCD, the caller's stream (sampling, mesh kernels, uncertainty) on the other %d" % (32 - _capi_mesh_cus(pipe), _capi_mesh_cus(pipe))
                               if pipe.mesh_stream is not None else ""))
                           if pipe._exclusive else
                           "encoder of step i+1 beside the head AND the mesh kernels of step i (small batch: neither fills the chip)" +
                           (", on disjoint CU subsets (encoder: %d of the 32 CUs of every XCD)" % (32 - _capi_mesh_cus(pipe)) if pipe.mesh_stream is not None else ", sharing all CUs"))},
            # The mesh kernel is FUSED (blend GEMM tile skinned in the MFMA epilogue, no v_posed round trip).  SURVEY 8(d):
            # it is still reported against the UNFUSED LBS definition (166,896 B per mesh), over the fused kernel's
            # whole launch time -- which also contains the 8.97 MFLOP/mesh blend GEMM that really bounds it (roofline_mfma).
            "roofline": {"bound": "hbm", "kernel": mesh_kernel, "fused": fused,
                         "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": (achieved / HBM_PEAK_GBS) if achieved else None, "traffic": traffic,
                         "traffic_imported": traffic_imported,      # False: measured in this run (live_traffic); True: read from the committed PMC summary
                         "traffic_source": traffic_source if traffic else None,
                         "traffic_detail": traffic_detail,
                         "avg_launch_ms": lbs_avg_ms, "launches": len(lbs_ms), "launch_spread": spread(lbs_ms),
                         "algorithmic_bytes_per_launch": LBS_BYTES_PER_MESH * M,
                         "note": ("fused kernel: the launch time covers blend GEMM + skinning; the bytes are SURVEY 8(d)'s "
                                  "unfused LBS definition (v_posed read + A read + verts write), of which only the verts "
                                  "write still exists" if fused else "unfused hps_smpl_lbs launch")},
            "roofline_mfma": ({"bound": "mfma", "kernel": mesh_kernel, "achieved": BLEND_FLOP_PER_MESH * M / (lbs_avg_ms * 1e-3) / 1e12,
                               "peak": MFMA_FP32_PEAK_TF, "unit": "TFLOP/s",
                               "frac": BLEND_FLOP_PER_MESH * M / (lbs_avg_ms * 1e-3) / 1e12 / MFMA_FP32_PEAK_TF,
                               "algorithmic_flop_per_launch": BLEND_FLOP_PER_MESH * M,
                               "k_algorithmic": 217, "k_required": k_required, "k_issued": 208 if shared_shape else 218,
                               "required_flop_per_launch": 2 * k_required * 3 * 6890 * M,
                               "frac_of_required_flop": 2 * k_required * 3 * 6890 * M / (lbs_avg_ms * 1e-3) / 1e12 / MFMA_FP32_PEAK_TF,
                               "note": "fp32 MFMA: the bound of the fused kernel.  achieved / frac use SURVEY 8(d)'s per-mesh figure "
                                       "(K = 217: pose blend + shape blend per mesh).  With use_mean_shape the meshes of an image share "
                                       "their betas and the shape blend is formed once per image (hps_smpl_v_shaped), so the launch "
                                       "has to multiply k_required = 207 rows per mesh (k_issued incl. padding): frac_of_required_flop "
                                       "is the fraction of the MFMA peak by the work the launch really does.  fp32 VALU work of the "
                                       "skinning epilogue does not overlap with fp32 MFMA on gfx950 (tools/mfma_valu_overlap.hip)"}
                              if fused and lbs_ms else None),
            "secondary": secondary,
            "metric_checksums": {"images": float(total[0]), "sum_unc": float(total[1]),
                                 "sum_abs_verts_mode": float(total[2]), "sum_abs_joints_samples": float(total[3])},
            # per-rank accumulators as gathered by the one collective (rank order); floats print with repr precision, so
            # equal strings mean equal bits
            "metric_checksums_per_rank": [[float(v) for v in row] for row in per_rank.tolist()],
            "backend": (torch.distributed.get_backend() if distributed else None),
            "image_range_rank0": [lo, hi],
        }
        if args.cpu_images > 0:
            # every N (north_star: "images/sec at 1/2/4/8 GPUs reported against the CPU reference baseline"): rank 0 times the oracle on
            # the host after the collective; the other ranks sleep in sharding.wait_for_rank0 below (no spinning: the host is rank 0's)
            out["cpu_baseline"] = cpu_baseline(net_state, configs.SMPL_PARENTS, N, args.cpu_images)
            out["speedup_vs_cpu_baseline"] = out["value"] / out["cpu_baseline"]["value"]
        print(json.dumps(out), flush=True)
    if distributed:
        sharding.wait_for_rank0("bench_line_printed", timeout_s=1800)
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
