"""GPU parity, end to end: proxy representation -> sampled meshes and uncertainty, HIP path vs CPU oracle on
the same seeded inputs (sample_on_cpu route = the reference's seed-reproducible route), BASELINE configs[0]
(B=1, N=1) and a small batch; full-size BASELINE configs[1] (B=64, N=100) through properties.

Stated tolerance: image -> vertices, joints, uncertainty <= 1e-4 m end to end (SURVEY.md section 8(c) allows 1e-3 m; DESIGN.md
claims 1e-4 and the tests hold it to that); observed ~6e-6."""
import pytest
import torch

from oracle import ref_cpu as O
from hierarchicalprobabilistic3dhuman_amd import _capi, configs, sharding
from hierarchicalprobabilistic3dhuman_amd.predict_poseMF_shapeGaussian_net import infer
from hierarchicalprobabilistic3dhuman_amd import sampling_utils as su
from hierarchicalprobabilistic3dhuman_amd import rigid_transform_utils as rtu
from conftest import maxerr

pytestmark = pytest.mark.gpu
KEYS = ("pose_F", "pose_S", "pose_rotmats_mode", "shape_loc", "shape_scale", "glob", "cam", "glob_rotmats",
        "verts_mode", "joints_mode", "verts_tpose", "R_samples", "verts_samples", "joints_samples", "unc")


@pytest.mark.parametrize("B,N", [(1, 1), (2, 4)])
def test_infer_matches_oracle(B, N, dev, net_gpu, net_cpu, smpl_gpu, smpl_assets, golden_input):
    x = golden_input[:B]
    torch.manual_seed(11)
    ref = O.infer(net_cpu[1], smpl_assets[2], configs.SMPL_PARENTS, x, N)
    torch.manual_seed(11)
    out = infer(net_gpu, smpl_gpu, x.to(dev), num_samples=N, sample_on_cpu=True)
    for k in KEYS:
        assert out[k].shape == ref[k].shape, k
        assert maxerr(out[k], ref[k]) <= 1e-4, k


def _oracle_infer_chunked(sd, params, x, N, chunk):
    """O.infer over ``chunk`` images at a time (the oracle materialises smplx's per-vertex 4x4 transforms: 441 KB per mesh).
    With use_mean_shape the only random draws are the sampler's, image by image, so the chunks continue one stream."""
    parts = [O.infer(sd, params, configs.SMPL_PARENTS, x[i:i + chunk], N) for i in range(0, x.shape[0], chunk)]
    return {k: torch.cat([p[k] for p in parts]) for k in KEYS + ("pose_U", "pose_V")}


def _assert_matches_oracle(out, ref, B, N, tol=1e-4):
    """All KEYS <= tol against the oracle, with the two DISCONTINUITIES of the reference's own function handled explicitly
    (counted and bounded, never silently skipped):

    (1) torch.svd's choice of signs for a singular-vector pair is not continuous in F: F_gpu and F_oracle differ by ~2e-7
        (accumulation order of the encoder), and for about one matrix in a thousand LAPACK then returns a differently signed
        pair (observed at configs[1]: image 32, joint 0, max |dF| = 2.4e-7, columns 2 and 3 of U and V negated).  U_proper feeds
        the children (models/poseMF_shapeGaussian_net.py:126-130), so that image's descendants, samples and meshes differ --
        exactly as they would between two runs of the reference on hosts whose convolutions round differently.  What IS
        required of such an image: the device's (U, S, V) equal torch.svd of the device's own F bit for bit (we ARE the
        reference's function at that input), checked for every matrix; everything that does not depend on the pose chain still
        matches; and at most max(1, B / 32) images are affected.
    (2) an accept decision on an fp32 rounding tie may flip (stated bound: <= 1e-6 of the proposals); a flip changes
        every later sample of that (image, joint) call."""
    o = {k: out[k].cpu() for k in KEYS + ("pose_U", "pose_V")}
    if _capi.svd_flavor_is_exact():
        Uh, Sh, Vh = torch.svd(o["pose_F"].reshape(-1, 3, 3))
        assert torch.equal(Uh.view_as(o["pose_U"]), o["pose_U"]) and torch.equal(Vh.view_as(o["pose_V"]), o["pose_V"])
        assert torch.equal(Sh.view_as(o["pose_S"]), o["pose_S"])
    dF = (o["pose_F"] - ref["pose_F"]).abs().amax(dim=(2, 3))
    dUV = torch.maximum((o["pose_U"] - ref["pose_U"]).abs().amax(dim=(2, 3)), (o["pose_V"] - ref["pose_V"]).abs().amax(dim=(2, 3)))
    split = (dF <= tol) & (dUV > 0.5)                  # same F, differently signed vector pair (ill-conditioning gives <= 1e-2)
    split_imgs = split.any(dim=1)
    assert int(split_imgs.sum()) <= max(1, B // 32), "LAPACK sign choice differs in %d images" % int(split_imgs.sum())
    good = (~split_imgs).nonzero().flatten()
    err = (o["R_samples"] - ref["R_samples"]).abs().amax(dim=(1, 3, 4))[good]            # (good images, 23)
    flipped = err > tol
    allowed = int(-(-B * 23 * 8 * N // 1000000))
    assert int(flipped.sum()) <= allowed, "accept decisions flipped in %d calls (allowed %d)" % (int(flipped.sum()), allowed)
    good_s = good[~flipped.any(dim=1)]
    pose_free = ("shape_loc", "shape_scale", "glob", "cam", "glob_rotmats", "verts_tpose")
    for k in KEYS:
        assert o[k].shape == ref[k].shape, k
        sel = slice(None) if k in pose_free else (good_s if k in ("R_samples", "verts_samples", "joints_samples", "unc") else good)
        assert maxerr(o[k][sel], ref[k][sel]) <= tol, k
    return int(split_imgs.sum()), int(flipped.sum())


def test_infer_matches_oracle_at_1000_samples(dev, net_gpu, net_cpu, smpl_gpu, smpl_assets, golden_input):
    """N = 1000: eight wavefronts per sampler call, the one-sweep uncertainty kernel, 2 004 meshes -- against the oracle."""
    torch.manual_seed(12)
    ref = _oracle_infer_chunked(net_cpu[1], smpl_assets[2], golden_input, 1000, 1)
    torch.manual_seed(12)
    out = infer(net_gpu, smpl_gpu, golden_input.to(dev), num_samples=1000, sample_on_cpu=True)
    _assert_matches_oracle(out, ref, 2, 1000)


@pytest.mark.parametrize("B,N,chunk", [(64, 100, 64), (16, 1000, 4)])
def test_full_size_configs_match_oracle(B, N, chunk, dev, net_gpu, net_cpu, smpl_gpu, smpl_assets):
    """BASELINE configs[1] (B = 64, N = 100: 6 528 meshes) and configs[4] (B = 16, N = 1000: 16 032 meshes) at their REAL
    size against the CPU oracle on the reference's seed-reproducible route (sample_on_cpu=True), all fifteen outputs <= 1e-4 --
    once through infer() and once through InferencePipeline (three streams; for configs[4] the CU-partition schedule)."""
    from hierarchicalprobabilistic3dhuman_amd.predict_poseMF_shapeGaussian_net import InferencePipeline
    x = torch.stack([torch.rand(18, 256, 256, generator=torch.Generator().manual_seed(5000 + i)) for i in range(B)])
    torch.manual_seed(13)
    ref = _oracle_infer_chunked(net_cpu[1], smpl_assets[2], x, N, chunk)
    xd = x.to(dev)
    torch.manual_seed(13)
    out = infer(net_gpu, smpl_gpu, xd, num_samples=N, sample_on_cpu=True)
    _assert_matches_oracle(out, ref, B, N)
    out = {k: out[k].cpu() for k in KEYS}
    pipe = InferencePipeline(net_gpu, smpl_gpu, num_samples=N, sample_on_cpu=True)
    with torch.cuda.stream(pipe.caller_stream(B)):
        torch.manual_seed(13)
        got = pipe.finish(pipe.submit(xd))
        got = {k: got[k].cpu() for k in KEYS}
    torch.cuda.synchronize()
    for k in KEYS:
        assert torch.equal(got[k], out[k]), k


@pytest.mark.parametrize("B,N,chunk", [(3, 7, 3), (5, 129, 5), (33, 20, 11)])
def test_ragged_sizes_match_oracle(B, N, chunk, dev, net_gpu, net_cpu, smpl_gpu, smpl_assets):
    """Sizes that divide nothing: 3 / 5 / 33 images (head batch tiles of four, odd encoder batches), 27 / 655 / 726 meshes (ragged
    64-mesh tiles of the fused kernel), 7 / 129 / 20 samples (one wavefront, the first size that takes two, a partial one) --
    every output against the oracle on the seed-reproducible route."""
    x = torch.stack([torch.rand(18, 256, 256, generator=torch.Generator().manual_seed(7000 + 31 * B + i)) for i in range(B)])
    torch.manual_seed(14)
    ref = _oracle_infer_chunked(net_cpu[1], smpl_assets[2], x, N, chunk)
    torch.manual_seed(14)
    out = infer(net_gpu, smpl_gpu, x.to(dev), num_samples=N, sample_on_cpu=True)
    _assert_matches_oracle(out, ref, B, N)


def test_reference_call_sequence_batch_one(dev, net_gpu, net_cpu, smpl_gpu, smpl_assets, golden_input):
    """The calls of predict/predict_poseMF_shapeGaussian_net.py:103-165, written as the reference writes them."""
    x = golden_input[:1].to(dev)
    pose_F, pose_U, pose_S, pose_V, mode, shape_dist, glob, cam = net_gpu(x)
    glob_rotmats = rtu.rot6d_to_rotmat(glob)
    out_mode = smpl_gpu(body_pose=mode, global_orient=glob_rotmats.unsqueeze(1), betas=shape_dist.loc, pose2rot=False)
    out_rest = smpl_gpu(betas=shape_dist.loc)
    torch.manual_seed(2)
    unc, verts, joints = su.compute_vertex_uncertainties_by_poseMF_shapeGaussian_sampling(
        pose_U=pose_U, pose_S=pose_S, pose_V=pose_V, shape_distribution=shape_dist, glob_rotmats=glob_rotmats,
        num_samples=6, smpl_model=smpl_gpu, use_mean_shape=True, sample_on_cpu=True)
    assert unc.shape == (6890,) and verts.shape == (6, 6890, 3) and joints.shape == (6, 90, 3)
    torch.manual_seed(2)
    ref = O.infer(net_cpu[1], smpl_assets[2], configs.SMPL_PARENTS, golden_input[:1], 6)
    assert maxerr(out_mode.vertices, ref["verts_mode"]) <= 1e-4 and maxerr(out_rest.vertices, ref["verts_tpose"]) <= 2e-5
    assert maxerr(verts, ref["verts_samples"][0]) <= 1e-4 and maxerr(unc, ref["unc"][0]) <= 1e-4


def test_full_size_config_properties(dev, net_gpu, smpl_gpu):
    """BASELINE configs[1]: B=64, N=100 on one GPU (6528 meshes): finite outputs, batch independence and
    sharding invariance of per-image outputs and of the reduced metric sums."""
    B, N = 64, 100
    x = torch.stack([torch.rand(18, 256, 256, generator=torch.Generator().manual_seed(1000 + i)) for i in range(B)]).to(dev)
    whole = infer(net_gpu, smpl_gpu, x, num_samples=N, seed=5)
    assert whole["verts_samples"].shape == (B, N, 6890, 3) and whole["unc"].shape == (B, 6890)
    for k in KEYS:
        assert torch.isfinite(whole[k]).all(), k
    R = whole["R_samples"]
    assert float((torch.matmul(R.transpose(-1, -2), R) - torch.eye(3, device=dev)).abs().max()) <= 1e-5
    assert float(whole["unc"].min()) >= 0.0
    # two "ranks" of 32 images reproduce the per-image outputs and the gathered metric sums
    parts = [infer(net_gpu, smpl_gpu, x[r * 32:(r + 1) * 32], num_samples=N, seed=5, image_offset=r * 32) for r in range(2)]
    for k in ("R_samples", "verts_mode", "unc"):
        assert maxerr(torch.cat([p[k] for p in parts]), whole[k]) <= 1e-5, k
    assert torch.equal(torch.cat([p["R_samples"] for p in parts]), whole["R_samples"]) or \
        maxerr(torch.cat([p["R_samples"] for p in parts]), whole["R_samples"]) <= 1e-5
    s_whole = sharding.batch_metric_sums(whole)
    s_parts = sharding.batch_metric_sums(parts[0]) + sharding.batch_metric_sums(parts[1])
    assert float(s_whole[0]) == 64.0 and maxerr(s_parts, s_whole) <= 1e-6 * float(s_whole.abs().max())


def test_pipelined_steps_equal_sequential_infer(dev, net_gpu, smpl_gpu, golden_input):
    """InferencePipeline (encoder of the next batch on a side stream) returns exactly what infer() returns."""
    from hierarchicalprobabilistic3dhuman_amd.predict_poseMF_shapeGaussian_net import InferencePipeline
    xs = [golden_input.to(dev), golden_input.flip(0).contiguous().to(dev), (golden_input * 0.5).to(dev)]
    want = [infer(net_gpu, smpl_gpu, x, num_samples=5, seed=40 + i) for i, x in enumerate(xs)]
    pipe = InferencePipeline(net_gpu, smpl_gpu, num_samples=5)
    got = []
    t = pipe.submit(xs[0])
    for i in range(len(xs)):
        nxt = pipe.submit(xs[i + 1]) if i + 1 < len(xs) else None
        got.append(pipe.finish(t, seed=40 + i, after=nxt))
        t = nxt
    torch.cuda.synchronize()
    for w, g in zip(want, got):
        for k in ("pose_F", "R_samples", "verts_samples", "unc", "verts_mode"):
            assert torch.equal(w[k], g[k]), k


@pytest.mark.parametrize("B,N", [(2, 8), (3, 50), (5, 128), (64, 100), (2, 129), (1, 5)])
def test_joint_regression_rides_in_the_uncertainty_launch(B, N, dev, net_gpu, smpl_gpu):
    """hps_joints_and_uncertainty (one launch: the joint regression of every mesh of the call + the uncertainty pass over its sample meshes)
    gives the bits of hps_smpl_joints + hps_vertex_uncertainty; sample counts outside the register-resident pass (N < 8, N > 128) keep the
    two launches."""
    from hierarchicalprobabilistic3dhuman_amd import predict_poseMF_shapeGaussian_net as pm
    calls = []
    real = _capi.call
    feats = (torch.rand(B, 512, generator=torch.Generator().manual_seed(B * 1000 + N)) * 2).to(dev)
    spy = lambda name, *a: (calls.append(name), real(name, *a))[1]
    try:
        _capi.call = spy
        fused = infer(net_gpu, smpl_gpu, None, num_samples=N, seed=3, input_feats=feats)
        names_fused = list(calls)
        pm.FUSE_JOINTS_AND_UNCERTAINTY = False
        del calls[:]
        two = infer(net_gpu, smpl_gpu, None, num_samples=N, seed=3, input_feats=feats)
        names_two = list(calls)
    finally:
        _capi.call = real
        pm.FUSE_JOINTS_AND_UNCERTAINTY = True
    assert ("hps_joints_and_uncertainty" in names_fused) == (8 <= N <= 128)
    assert "hps_joints_and_uncertainty" not in names_two and "hps_smpl_joints" in names_two and "hps_vertex_uncertainty" in names_two
    for k in ("joints_mode", "joints_samples", "unc", "verts_samples"):
        assert torch.equal(fused[k], two[k]), k
    assert torch.isfinite(fused["unc"]).all() and torch.isfinite(fused["joints_samples"]).all()


def test_mesh_kernel_on_the_encoder_stream_equals_the_separate_stream_schedule(dev, net_gpu, smpl_gpu):
    """InferencePipeline.inline_mesh (the mesh kernel queued on the encoder's stream; its operands are NOT registered with that stream -- the
    caller's stream waits for the kernel instead, SMPL.forward "ordered") returns the bits of the schedule that keeps the kernel on the caller's
    stream, over steps that free and re-allocate every operand (the allocator reuses the blocks at once)."""
    from hierarchicalprobabilistic3dhuman_amd.predict_poseMF_shapeGaussian_net import InferencePipeline
    torch.manual_seed(5)
    xs = [torch.rand(32, 18, 256, 256, device=dev) for _ in range(2)]
    sums = {}
    # (inline_mesh, inline_side): mesh kernel + joints + uncertainty on the encoder's stream (the default), the mesh kernel alone there,
    # everything on the caller's stream
    for inline, side in ((True, True), (False, False), (True, False), (True, True)):
        pipe = InferencePipeline(net_gpu, smpl_gpu, num_samples=20)
        pipe.inline_mesh, pipe.inline_side = inline, side
        acc = []
        t = pipe.submit(xs[0], input_ready=False)
        for i in range(6):
            nxt = pipe.submit(xs[(i + 1) % 2], input_ready=False) if i < 5 else None
            res = pipe.finish(t, seed=90 + i, after=nxt)
            acc.append(torch.stack([res["verts_samples"].double().sum(), res["unc"].double().sum(), res["joints_samples"].double().sum()
                                    if "joints_samples" in res else res["verts_mode"].double().sum()]))
            scratch = torch.empty_like(res["verts_samples"]).fill_(float(i))      # churn: the freed blocks are taken again immediately
            del res, scratch
            t = nxt
        torch.cuda.synchronize()
        got = torch.stack(acc).cpu()
        assert torch.isfinite(got).all()
        sums.setdefault((inline, side), []).append(got)
    first = sums[(True, True)][0]
    assert all(torch.equal(first, g) for runs in sums.values() for g in runs)


def test_an_error_inside_the_mesh_window_leaves_the_caller_on_its_own_stream(dev, net_gpu, smpl_gpu, monkeypatch):
    """Between the pipeline's two hooks the current stream is the encoder's (inline_mesh); a launch that fails there must not leave it so."""
    from hierarchicalprobabilistic3dhuman_amd import _capi
    from hierarchicalprobabilistic3dhuman_amd.predict_poseMF_shapeGaussian_net import InferencePipeline
    pipe = InferencePipeline(net_gpu, smpl_gpu, num_samples=20)
    x = torch.rand(32, 18, 256, 256, device=dev)
    main = torch.cuda.current_stream()
    want = pipe.finish(pipe.submit(x, input_ready=False), seed=3)["verts_samples"].clone()
    ticket = pipe.submit(x, input_ready=False)
    real = _capi.call

    def failing(name, *args):
        if name.startswith("hps_smpl_mesh_fused"):
            assert torch.cuda.current_stream() != main           # (the case under test: the kernel is queued on the encoder's stream)
            raise _capi.HpsError("injected launch failure")
        return real(name, *args)

    monkeypatch.setattr(_capi, "call", failing)
    with pytest.raises(_capi.HpsError):
        pipe.finish(ticket, seed=3)
    assert torch.cuda.current_stream() == main
    monkeypatch.setattr(_capi, "call", real)
    got = pipe.finish(pipe.submit(x, input_ready=False), seed=3)["verts_samples"]      # and the pipeline still works
    torch.cuda.synchronize()
    assert torch.equal(got, want)


def test_pipeline_from_host_rgb_equals_infer_on_the_proxy_representation(dev, net_gpu, smpl_gpu):
    """The reference's order of work (predict/...:61-104) as the pipelined loop runs it: page-locked host RGB crops + keypoints ->
    StagedUpload (copy stream, two device slots) -> submit(make_input=...) builds the proxy representation on the encoder's stream
    -> finish().  Same bits as infer() on a proxy representation built up front, for every batch and slot reuse."""
    from hierarchicalprobabilistic3dhuman_amd import configs
    from hierarchicalprobabilistic3dhuman_amd.canny_edge_detector import CannyEdgeDetector
    from hierarchicalprobabilistic3dhuman_amd.predict_poseMF_shapeGaussian_net import (InferencePipeline, StagedUpload,
                                                                                        proxy_representation)
    cfg = configs.get_cfg_defaults()
    det = CannyEdgeDetector(cfg.DATA.EDGE_NMS, cfg.DATA.EDGE_GAUSSIAN_STD, cfg.DATA.EDGE_GAUSSIAN_SIZE, cfg.DATA.EDGE_THRESHOLD).to(dev)
    B, N = 3, 6
    host = []
    for k in range(5):
        g = torch.Generator().manual_seed(900 + k)
        rgb = torch.nn.functional.interpolate(torch.rand(B, 3, 32, 32, generator=g), size=(256, 256), mode="bilinear", align_corners=False)
        host.append([t.pin_memory() for t in (rgb, torch.rand(B, 17, 2, generator=g) * 256, (torch.rand(B, 17, generator=g) > 0.2).float())])
    want = []
    for k, (rgb, j, v) in enumerate(host):
        proxy = proxy_representation(rgb.to(dev), j.to(dev), v.to(dev), det, cfg)
        r = infer(net_gpu, smpl_gpu, proxy, num_samples=N, seed=60 + k)
        want.append({key: r[key].clone() for key in ("pose_F", "R_samples", "verts_mode", "unc")})
    pipe = InferencePipeline(net_gpu, smpl_gpu, num_samples=N)
    pipe.caller_stream(B)
    stager = StagedUpload(slots=2)

    # encoder=None: the (B,18,D,D) tensor is built and the encoder splits it into phase frames; encoder=...: the front end writes the
    # Winograd stem's phase frames directly (hps_proxy_rep_phase_frames, round 5) -- the same frames, so the same bits
    for enc in (None, net_gpu.image_encoder):
        def step(k):
            (rgb_d, j_d, v_d), ready = stager.upload(host[k])
            t = pipe.submit(make_input=lambda: proxy_representation(rgb_d, j_d, v_d, det, cfg, encoder=enc), input_ready=ready)
            stager.release(t[1])
            return t

        got = []
        t = step(0)
        for k in range(len(host)):
            nxt = step(k + 1) if k + 1 < len(host) else None
            r = pipe.finish(t, seed=60 + k, after=nxt)
            got.append({key: r[key].clone() for key in want[k]})
            t = nxt
        torch.cuda.synchronize()
        for w, g_ in zip(want, got):
            for key in w:
                assert torch.equal(w[key], g_[key]), (key, enc is not None)


def test_front_end_writes_the_stems_phase_frames_directly(dev, net_gpu):
    """VERDICT r4 item 3b.  hps_proxy_rep_phase_frames leaves in the encoder's phase-frame buffer exactly what hps_proxy_rep +
    hps_stem_phase_split leave there -- every float of the buffer, halo and padding included -- for visibility masks, joints outside
    the image and both edge-map selections; features from the filled frames equal features from the NCHW tensor; shapes the
    Winograd stem does not take fall back to the tensor."""
    from hierarchicalprobabilistic3dhuman_amd import configs
    from hierarchicalprobabilistic3dhuman_amd.canny_edge_detector import CannyEdgeDetector
    from hierarchicalprobabilistic3dhuman_amd.predict_poseMF_shapeGaussian_net import proxy_representation
    from hierarchicalprobabilistic3dhuman_amd.resnet import FilledStemFrames
    enc = net_gpu.image_encoder
    for B, D, nms in ((2, 256, True), (5, 256, False), (3, 64, True)):
        cfg = configs.get_cfg_defaults()
        cfg.DATA.PROXY_REP_SIZE, cfg.DATA.EDGE_NMS = D, nms
        det = CannyEdgeDetector(True, cfg.DATA.EDGE_GAUSSIAN_STD, cfg.DATA.EDGE_GAUSSIAN_SIZE, cfg.DATA.EDGE_THRESHOLD).to(dev)
        g = torch.Generator().manual_seed(B * 7 + D)
        rgb = torch.nn.functional.interpolate(torch.rand(B, 3, D // 8, D // 8, generator=g), size=(D, D), mode="bilinear", align_corners=False).to(dev)
        j2d = (torch.rand(B, 17, 2, generator=g) * 1.4 - 0.2) * D                      # some joints outside the image
        vis = (torch.rand(B, 17, generator=g) > 0.3).float()
        for v in (vis.to(dev), None):
            proxy = proxy_representation(rgb, j2d.to(dev), v, det, cfg)
            assert torch.is_tensor(proxy) and proxy.shape == (B, 18, D, D)
            want_feats = enc(proxy).clone()
            filled_buf = enc.stem_frames(B, 18, D, D, dev)
            _capi.call("hps_stem_phase_split", _capi.ptr(proxy), _capi.ptr(filled_buf.frames), B, 18, D, D, _capi.stream())
            want_frames = filled_buf.frames.clone()                                      # as hps_stem_phase_split leaves them
            filled_buf.frames.fill_(-9.0)                                                # prove every in-image float is rewritten ...
            filled = proxy_representation(rgb, j2d.to(dev), v, det, cfg, encoder=enc)
            assert isinstance(filled, FilledStemFrames) and filled.frames.data_ptr() == filled_buf.frames.data_ptr()
            assert filled.fill is not None and bool((filled.frames == -9.0).all())       # the fill is deferred to forward(): nothing written yet
            filled.run_fill()
            touched = filled.frames != -9.0
            assert torch.equal(filled.frames[touched], want_frames[touched])
            assert bool((want_frames[~touched] == 0).all())                              # ... and only the zero halo / slack is not
            filled.frames.copy_(torch.where(touched, filled.frames, torch.zeros_like(filled.frames)))
            assert torch.equal(filled.frames, want_frames)
            assert torch.equal(enc(filled), want_feats)
            with pytest.raises(_capi.HpsError, match="stale"):                           # one forward per stem_frames(): the buffer is shared
                enc(filled)
            with pytest.raises(_capi.HpsError, match="stale"):                           # ... and an object handed out earlier is stale too
                enc(filled_buf)
    # a shape the Winograd stem does not take: the tensor route
    cfg.DATA.PROXY_REP_SIZE = 48
    rgb = torch.rand(2, 3, 48, 48).to(dev)
    out = proxy_representation(rgb, (torch.rand(2, 17, 2) * 48).to(dev), None, det, cfg, encoder=enc)
    assert torch.is_tensor(out) and out.shape == (2, 18, 48, 48)


def test_predict_loop_from_host_proxies_equals_infer(dev, net_gpu, smpl_gpu, tmp_path):
    """predict_poseMF_shapeGaussian_net (reference signature, run_predict.py:77-89) with per-image proxy representations in HOST
    memory (page-locked and pageable): the loop stages them over two copy streams into alternating device slots and
    software-pipelines the batches -- every image's outputs must be those of infer() on its batch, bit for bit (7 images in
    batches of 3: a ragged last batch; slots reused)."""
    import os
    from hierarchicalprobabilistic3dhuman_amd import configs
    from hierarchicalprobabilistic3dhuman_amd.predict_poseMF_shapeGaussian_net import predict_poseMF_shapeGaussian_net
    cfg = configs.get_cfg_defaults()
    names = ["img_%02d.png" % i for i in range(7)]
    image_dir = tmp_path / "images"
    os.makedirs(image_dir)
    for n in names:
        open(image_dir / n, "wb").close()                                  # names only: proxy_rep_fn supplies the content
    proxies = {n: torch.rand(1, 18, 256, 256, generator=torch.Generator().manual_seed(700 + i)) for i, n in enumerate(names)}
    for pinned in (True, False):
        store = {n: (t.pin_memory() if pinned else t) for n, t in proxies.items()}
        got = {}
        torch.manual_seed(21)
        predict_poseMF_shapeGaussian_net(net_gpu, cfg, smpl_gpu, None, None, None, dev, str(image_dir), str(tmp_path / "out"),
                                         proxy_rep_fn=lambda path: store[os.path.basename(path)], num_samples=5, batch_size=3,
                                         result_fn=lambda name, item: got.__setitem__(name, {k: item[k].clone() for k in ("pose_F", "R_samples", "verts_mode", "unc")}))
        torch.cuda.synchronize()
        assert sorted(got) == names
        torch.manual_seed(21)                                              # the loop draws one Philox seed per batch from this generator
        for i0 in range(0, len(names), 3):
            batch = names[i0:i0 + 3]
            want = infer(net_gpu, smpl_gpu, torch.cat([proxies[n] for n in batch]).to(dev), num_samples=5)
            for k, n in enumerate(batch):
                for key in got[n]:
                    assert torch.equal(got[n][key], want[key][k]), (pinned, n, key)


@pytest.mark.parametrize("batch_size", [1, 2])
def test_predict_loop_one_image_at_a_time_runs_on_graphs_and_equals_infer(batch_size, dev, net_gpu, smpl_gpu, tmp_path):
    """The reference's own operating point (predict/...:58-59: one image per call): predict_poseMF_shapeGaussian_net with batch_size 1 / 2
    replays hipGraphs (GraphedInfer, two slots) -- every image's outputs are those of infer() with the same seed, bit for bit, the
    callback's tensors stay valid after later replays, a smaller last group takes the eager path, in both encoder modes."""
    import os
    from hierarchicalprobabilistic3dhuman_amd import configs
    from hierarchicalprobabilistic3dhuman_amd.predict_poseMF_shapeGaussian_net import predict_poseMF_shapeGaussian_net
    cfg = configs.get_cfg_defaults()
    names = ["img_%02d.png" % i for i in range(5)]
    image_dir = tmp_path / "images"
    os.makedirs(image_dir)
    for n in names:
        open(image_dir / n, "wb").close()
    proxies = {n: torch.rand(1, 18, 256, 256, generator=torch.Generator().manual_seed(900 + i)).pin_memory() for i, n in enumerate(names)}
    keys = ("pose_F", "R_samples", "verts_mode", "verts_samples", "joints_samples", "unc")
    try:
        for latency in (False, True):
            net_gpu.set_latency_mode(latency)
            got = {}
            torch.manual_seed(31)
            predict_poseMF_shapeGaussian_net(net_gpu, cfg, smpl_gpu, None, None, None, dev, str(image_dir), str(tmp_path / "out"),
                                             proxy_rep_fn=lambda path: proxies[os.path.basename(path)], num_samples=6, batch_size=batch_size,
                                             result_fn=lambda name, item: got.__setitem__(name, {k: item[k] for k in keys}))      # kept, NOT cloned
            torch.cuda.synchronize()
            assert sorted(got) == names
            torch.manual_seed(31)
            for i0 in range(0, len(names), batch_size):
                batch = names[i0:i0 + batch_size]
                want = infer(net_gpu, smpl_gpu, torch.cat([proxies[n] for n in batch]).to(dev), num_samples=6)
                for k, n in enumerate(batch):
                    for key in keys:
                        assert torch.equal(got[n][key], want[key][k]), (latency, n, key)
    finally:
        net_gpu.set_latency_mode(False)


def test_graphed_infer_equals_infer_and_draws_new_samples_on_every_replay(dev, net_gpu, smpl_gpu, golden_input):
    """GraphedInfer: the captured launches are infer()'s (same bits for the same seed / image offset), the Philox key is read at run time
    (other seeds -> other samples, the same seed -> the same samples, replay after replay), slots alternate, and a wrong input shape is
    refused."""
    from hierarchicalprobabilistic3dhuman_amd.predict_poseMF_shapeGaussian_net import GraphedInfer
    x = golden_input.to(dev)
    g = GraphedInfer(net_gpu, smpl_gpu, batch=2, num_samples=7, slots=2)
    keys = ("pose_F", "pose_rotmats_mode", "R_samples", "verts_mode", "verts_tpose", "verts_samples", "joints_samples", "unc", "cam")
    outs = []
    for seed, off in ((5, 0), (6, 0), (5, 0), (5, 64), (6, 0)):
        want = infer(net_gpu, smpl_gpu, x, num_samples=7, seed=seed, image_offset=off)
        got = g(x, seed=seed, image_offset=off)
        for k in keys:
            assert torch.equal(got[k], want[k]), (seed, off, k)
        outs.append(got["R_samples"].clone())
    assert torch.equal(outs[0], outs[2]) and torch.equal(outs[1], outs[4])
    assert not torch.equal(outs[0], outs[1]) and not torch.equal(outs[0], outs[3])
    g.check_sampling()
    with pytest.raises(_capi.HpsError, match="captured for inputs"):
        g(x[:1])
    # two replays in flight: issue both, then wait
    (o1, d1), (o2, d2) = g(x, seed=11, wait=False), g(x.flip(0).contiguous(), seed=12, wait=False)
    d1.synchronize(); d2.synchronize()
    w1 = infer(net_gpu, smpl_gpu, x, num_samples=7, seed=11)
    w2 = infer(net_gpu, smpl_gpu, x.flip(0).contiguous(), num_samples=7, seed=12)
    assert torch.equal(o1["verts_samples"], w1["verts_samples"]) and torch.equal(o2["verts_samples"], w2["verts_samples"])


def test_bench_line_has_every_leg(dev):
    """The line the driver records: one short run of bench.py with all of its legs on (headline, lbs_unfused, latency_b1,
    from_rgb), every field the contract names present and consistent."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--steps", "3", "--warmup", "2", "--cpu-images", "1", "--lbs-unfused-reps", "2",
           "--latency-reps", "3", "--from-rgb-steps", "8", "--stress-steps", "3"]
    p = subprocess.run(cmd, cwd=root, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert p.returncode == 0, p.stderr.decode(errors="replace")[-3000:]
    lines = [l for l in p.stdout.decode().splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["dtype"] == "f32" and d["vs_baseline"] is None
    assert d["config"]["workload"].startswith("BASELINE configs[1]") and d["config"]["input_sets"] == 2
    assert abs(d["value"] - 64 * 3 / (d["ms_per_step"] * 3e-3)) <= 1e-6 * d["value"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["fused"] and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    # roofline.traffic is MEASURED IN THIS RUN where rocprofv3 exists (two --pmc passes of a three-step sub-run, bench.live_traffic), else
    # read from the committed PMC summary and flagged; either way it is the fused kernel's fabric-side bytes: at least the vertices it
    # writes (540 MB at 6 528 meshes), well below twice the unfused LBS definition's bytes
    import shutil
    if shutil.which("rocprofv3") is not None:
        assert r["traffic_imported"] is False and r["traffic_detail"]["launches"] >= 2, (r["traffic_imported"], r["traffic_source"])
    assert r["traffic"] is None or (r["traffic_imported"] in (True, False) and 5.3e8 <= r["traffic"] <= 2 * r["algorithmic_bytes_per_launch"])
    sec = d["secondary"]
    assert sec["latency_b1"]["median_ms"] > 0 and sec["latency_b1"]["throughput_mode_median_ms"] > 0
    assert 0 < sec["latency_b1"]["graph_median_ms"] <= 1.2 * sec["latency_b1"]["median_ms"]          # a replay is not slower than issuing the launches
    assert sec["latency_b1"]["throughput_mode_graph_median_ms"] > 0
    assert sec["from_rgb"]["images_per_s"] > 0 and len(sec["from_rgb"]["legs_images_per_s"]) == 3
    assert sec["from_rgb"]["checksum_images"] == 64 * 8
    # the three legs of the PCIe-inclusive loop must agree: a leg that is still warming is not a measurement
    legs = sec["from_rgb"]["legs_images_per_s"]
    assert (max(legs) - min(legs)) / max(legs) <= 0.10, legs
    # BASELINE configs[4] and the reference's own operating point on the CPU ride in the same line
    st = sec["stress_n1000"]
    assert st["workload"].startswith("BASELINE configs[4]") and st["images_per_s"] > 0 and st["results_finite"]
    assert st["sampler"]["proposals_per_s"] > 0 and 0 < st["mesh_kernel"]["frac"] < 1
    assert 0 < st["uncertainty_sweep1"]["frac"] < 1 and 0 < st["lbs_unfused"]["frac"] < 1
    # the opt-in bf16x3 arithmetic of the mesh kernel under the same clock: faster kernel, vertices within rounding of the fp32-MFMA kernel's
    mb = sec["mesh_bf16x3"]
    assert mb["images_per_s"] > 0 and mb["mesh_kernel_ms"]["median_ms"] < r["avg_launch_ms"]
    assert 0 < mb["max_abs_diff_vs_f32_m"]["verts_samples"] <= 4e-6 and mb["max_abs_diff_vs_f32_m"]["vertex_uncertainty"] <= 4e-6
    assert st["mesh_bf16x3"]["images_per_s"] > 0 and d["config"]["mesh_arith"] == "f32"
    cl = d["cpu_baseline"]["latency_b1"]
    assert cl["all_threads"]["median_ms"] > 0 and cl["single_thread"]["cores"] == 1
    assert sec["lbs_unfused"]["frac"] > 0 and d["cpu_baseline"]["kind"] == "port" and d["metric_checksums"]["images"] == 64 * 3


def test_pipeline_on_cu_partitions_equals_sequential_infer(dev, net_gpu, smpl_gpu):
    """Small batches with a long mesh chain (BASELINE configs[4]: 16 images x 1000 samples = 16 032 meshes) run the encoder and
    the mesh kernels side by side on disjoint CU subsets (hps_stream_create_cu_partition, chosen automatically); a batch with a
    tenth of the meshes keeps sharing the CUs.  Same bits as infer() either way; the loop runs on the stream caller_stream names."""
    from hierarchicalprobabilistic3dhuman_amd.predict_poseMF_shapeGaussian_net import InferencePipeline
    B, N = 16, 1000
    xs = [torch.stack([torch.rand(18, 256, 256, generator=torch.Generator().manual_seed(3000 + 16 * j + i)) for i in range(B)]).to(dev)
          for j in range(2)]
    want = [infer(net_gpu, smpl_gpu, x, num_samples=N, seed=70 + i) for i, x in enumerate(xs)]
    want = [{k: w[k].clone() for k in ("pose_F", "R_samples", "unc", "verts_mode")} for w in want]
    pipe = InferencePipeline(net_gpu, smpl_gpu, num_samples=N)
    loop = pipe.caller_stream(B)
    assert pipe.mesh_stream is not None and loop == pipe.mesh_stream and not pipe._exclusive        # the partition is in use
    torch.cuda.synchronize()
    got = []
    with torch.cuda.stream(loop):
        t = pipe.submit(xs[0])
        for i in range(len(xs)):
            nxt = pipe.submit(xs[i + 1]) if i + 1 < len(xs) else None
            r = pipe.finish(t, seed=70 + i, after=nxt)
            got.append({k: r[k].clone() for k in ("pose_F", "R_samples", "unc", "verts_mode")})
            t = nxt
    torch.cuda.synchronize()
    for w, g in zip(want, got):
        for k in w:
            assert torch.equal(w[k], g[k]), k
    small = InferencePipeline(net_gpu, smpl_gpu, num_samples=100)
    small.caller_stream(B)
    assert small.mesh_stream is None and not small._exclusive                    # shared CUs, mesh kernel beside the encoder
    big = InferencePipeline(net_gpu, smpl_gpu, num_samples=100)
    big.caller_stream(64)
    assert big.mesh_stream is None and big._exclusive                            # B = 64: the mesh kernel takes turns with the encoders


def test_stress_config_1000_samples_per_image(dev, net_gpu, smpl_gpu):
    """BASELINE configs[4]: num_samples=1000 on one GPU (B=16 -> 16 032 meshes, 1.3 GB of vertices): properties only."""
    B, N = 16, 1000
    x = torch.stack([torch.rand(18, 256, 256, generator=torch.Generator().manual_seed(2000 + i)) for i in range(B)]).to(dev)
    out = infer(net_gpu, smpl_gpu, x, num_samples=N, seed=9)
    assert out["verts_samples"].shape == (B, N, 6890, 3) and out["joints_samples"].shape == (B, N, 90, 3)
    assert torch.isfinite(out["verts_samples"]).all() and torch.isfinite(out["unc"]).all()
    R = out["R_samples"]
    assert float((torch.matmul(R.transpose(-1, -2), R) - torch.eye(3, device=dev)).abs().max()) <= 1e-5
    # sample 0..99 of image i equal a 100-sample run?  No (different N changes the proposal stream length), but the
    # per-image outputs must not depend on the batch: image 5 alone reproduces its slice bit for bit
    solo = infer(net_gpu, smpl_gpu, x[5:6], num_samples=N, seed=9, image_offset=5)
    assert torch.equal(solo["R_samples"][0], out["R_samples"][5])
    assert maxerr(solo["verts_samples"][0], out["verts_samples"][5]) <= 1e-5
    assert maxerr(solo["unc"][0], out["unc"][5]) <= 1e-5
    # the uncertainty of the mean shape under many samples is smooth and strictly positive on a posed body
    assert float(out["unc"].min()) > 0.0


def test_result_checksums_match_float64_sums(dev):
    """hps_sums_f64 (the accumulator bench.py and the sharding tests use): float64 sums in a fixed order, repeatable bit for
    bit, equal to torch's float64 reductions to rounding."""
    g = torch.Generator().manual_seed(3)
    res = {"unc": torch.rand(5, 6890, generator=g).to(dev), "verts_mode": torch.randn(5, 6890, 3, generator=g).to(dev),
           "joints_samples": torch.randn(5, 7, 90, 3, generator=g).to(dev)}
    a = sharding.batch_metric_sums(res)
    b = sharding.batch_metric_sums(res)
    assert a.dtype == torch.float64 and torch.equal(a, b)
    want = torch.stack([torch.tensor(5.0, dtype=torch.float64, device=dev), res["unc"].double().sum(),
                        res["verts_mode"].double().abs().sum(), res["joints_samples"].double().abs().sum()])
    assert maxerr(a, want) <= 1e-12 * float(want.abs().max())


def test_bf16x3_mesh_arithmetic_end_to_end(dev, net_gpu, net_cpu, smpl_gpu, smpl_assets, golden_input):
    """SMPL.mesh_arith = "bf16x3" through infer(), InferencePipeline and GraphedInfer: the three agree bit for bit among themselves, the
    vertices stay within rounding (4e-6 m) of the default fp32-MFMA arithmetic and within the stated 1e-4 m of the oracle end to end, and
    everything in front of the mesh kernel (head, samples) is untouched."""
    from hierarchicalprobabilistic3dhuman_amd.predict_poseMF_shapeGaussian_net import InferencePipeline, GraphedInfer
    x = golden_input.to(dev)
    keys = ("pose_F", "R_samples", "verts_mode", "verts_tpose", "verts_samples", "joints_samples", "unc", "cam")
    ref32 = infer(net_gpu, smpl_gpu, x, num_samples=9, seed=3)
    smpl_gpu.mesh_arith = "bf16x3"
    try:
        a = infer(net_gpu, smpl_gpu, x, num_samples=9, seed=3)
        pipe = InferencePipeline(net_gpu, smpl_gpu, num_samples=9)
        b = pipe.finish(pipe.submit(x), seed=3)
        g = GraphedInfer(net_gpu, smpl_gpu, batch=2, num_samples=9, slots=1)
        c = g(x, seed=3)
        torch.cuda.synchronize()
        for k in keys:
            assert torch.equal(a[k], b[k]) and torch.equal(a[k], c[k]), k
        del pipe, g
    finally:
        smpl_gpu.mesh_arith = "f32"
    for k in ("pose_F", "R_samples", "cam"):
        assert torch.equal(a[k], ref32[k]), k
    for k in ("verts_mode", "verts_tpose", "verts_samples", "joints_samples", "unc"):
        assert 0 <= maxerr(a[k], ref32[k]) <= 4e-6, k
    assert not torch.equal(a["verts_samples"], ref32["verts_samples"])          # (the switch did select the other kernel)
    # against the oracle on the reference's seed-reproducible route (the host noise stream), as test_infer_matches_oracle does for "f32"
    torch.manual_seed(11)
    ref = O.infer(net_cpu[1], smpl_assets[2], configs.SMPL_PARENTS, golden_input, 4)
    smpl_gpu.mesh_arith = "bf16x3"
    try:
        torch.manual_seed(11)
        out = infer(net_gpu, smpl_gpu, x, num_samples=4, sample_on_cpu=True)
    finally:
        smpl_gpu.mesh_arith = "f32"
    for k in KEYS:
        assert maxerr(out[k], ref[k]) <= 1e-4, k


@pytest.mark.parametrize("B,N", [(64, 100), (16, 1000)])
def test_bf16x3_mesh_arithmetic_at_full_size(B, N, dev, net_gpu, smpl_gpu):
    """BASELINE configs[1] and configs[4] with SMPL.mesh_arith = "bf16x3": finite, within rounding (4e-6 m) of the default arithmetic on
    every vertex of every sample mesh, and a per-image output does not depend on the batch it was computed in -- an image alone, or a
    half batch at another image offset, reproduces its slice BIT FOR BIT (a mesh's tile, neighbours and template group change, its sums do
    not: the sharding invariance the multi-GPU path rests on)."""
    x = torch.stack([torch.rand(18, 256, 256, generator=torch.Generator().manual_seed(3000 + i)) for i in range(B)]).to(dev)
    ref = infer(net_gpu, smpl_gpu, x, num_samples=N, seed=21)
    smpl_gpu.mesh_arith = "bf16x3"
    try:
        out = infer(net_gpu, smpl_gpu, x, num_samples=N, seed=21)
        solo = infer(net_gpu, smpl_gpu, x[5:6], num_samples=N, seed=21, image_offset=5)
        half = infer(net_gpu, smpl_gpu, x[B // 2:], num_samples=N, seed=21, image_offset=B // 2)
    finally:
        smpl_gpu.mesh_arith = "f32"
    assert torch.equal(out["R_samples"], ref["R_samples"])
    for k in ("verts_samples", "verts_mode", "verts_tpose", "joints_samples", "unc"):
        assert torch.isfinite(out[k]).all(), k
        assert maxerr(out[k], ref[k]) <= 4e-6, k
    for k in ("verts_samples", "verts_mode", "verts_tpose", "joints_samples", "joints_mode", "unc"):
        assert torch.equal(solo[k][0], out[k][5]), k
        assert torch.equal(half[k], out[k][B // 2:]), k
