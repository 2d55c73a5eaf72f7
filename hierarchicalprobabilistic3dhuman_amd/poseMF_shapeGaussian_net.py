"""``PoseMFShapeGaussianNet`` with the constructor, state-dict layout and forward contract of the
reference's models/poseMF_shapeGaussian_net.py, executed by libhps.so.

    net = PoseMFShapeGaussianNet(smpl_parents=smpl.parents.tolist(), config=cfg).to(device).eval()
    net.load_state_dict(checkpoint['best_model_state_dict'])
    pose_F, pose_U, pose_S, pose_V, pose_rotmats_mode, shape_dist, glob, cam = net(proxy_rep_input)

Execution: ResNet-18 encoder (resnet.py) -> FC trunk (hps_linear) -> the 23 per-joint MLPs grouped into
kinematic depth levels (hps_head_joint_level; 8 levels for the SMPL tree instead of 23 sequential steps)
-> 3x3 SVD -> proper-SVD fix and mode (hps_head_svd_finish).

3x3 SVD (:137, "SVD is faster on CPU" in the reference).  The column signs LAPACK returns are not determined by the
mathematics, they feed the child joints' MLPs through U_proper (:126-130), and the trained weights were fitted to them
(SURVEY.md section 7 hard part 1).  Two modes, ``net.svd_mode``:
  "device" (default)  the SVD runs inside the level kernel and follows LAPACK's sgesdd step by step WITH MKL'S ROUNDINGS
                      (csrc/svd3_gesdd.h): U, S and V are bit-identical to torch.svd on this host (10^6 matrices of 22 families,
                      tests/test_host_logic.py).  MKL rounds differently on Intel and on other hosts (fused multiply-adds or
                      not -- one differently signed vector pair in 10^4 matrices between the two, i.e. the reference itself is
                      only reproducible across hosts to that level); ``svd_flavor`` = None takes the flavour of this host,
                      0 / 1 force the reference-BLAS / fused one.  The head is 11 stream-ordered launches, no host
                      synchronisation.
  "host"              the reference's very routine: MKL sgesdd on the host, one D2H / sync / H2D round trip per kinematic
                      level.
"""
import os

import torch
from torch import nn
from torch.distributions import Normal

from . import _capi
from .resnet import resnet18
from .rigid_transform_utils import rotmat_to_rot6d
from .sharding import effective_cpus


# host SVD pool size: half the usable hardware threads (cgroup quota aware), shared between the ranks torchrun started on this node
_SVD_THREADS = max(2, min(16, effective_cpus() // (2 * max(1, int(os.environ.get("LOCAL_WORLD_SIZE", "1"))))))


def _host_svd_packed(f_host, usv_host):
    """SVD of n 3x3 matrices on the host (f_host (n,3,3) -> usv_host (n,21) packed [U | S | V]) through
    hps_host_svd3_packed: the same MKL sgesdd_ torch.svd calls, bit-identical factors, but the independent
    matrices are spread over a small native thread pool (torch's batched CPU SVD is a sequential loop,
    ~1.7 us per matrix)."""
    n = f_host.shape[0]
    assert f_host.is_contiguous() and usv_host.is_contiguous() and not f_host.is_cuda
    _capi.call("hps_host_svd3_packed", _capi._P(f_host.data_ptr()), _capi._P(usv_host.data_ptr()), n, _SVD_THREADS)


def immediate_parents_to_all_parents(immediate_parents):
    """models/poseMF_shapeGaussian_net.py:14-21: for every body joint (SMPL joint - 1) the list of its
    ancestors, nearest first, the root excluded.  Returned as a plain dict joint -> list."""
    all_parents = {}
    for smpl_idx in range(1, len(immediate_parents)):
        joint, parent = smpl_idx - 1, immediate_parents[smpl_idx] - 1
        all_parents[joint] = [parent] + all_parents[parent] if parent >= 0 else []
    return all_parents


def _invalidate_after_load(module, incompatible_keys):
    """load_state_dict post hook (fires for sub-modules too); module level so that the module stays picklable."""
    module.invalidate()


class PoseMFShapeGaussianNet(nn.Module):
    def __init__(self, smpl_parents, config):
        super().__init__()
        self.config = config
        self.parents_dict = immediate_parents_to_all_parents(smpl_parents)
        self.num_joints = len(self.parents_dict)
        self.num_pose_params = self.num_joints * 9
        self.num_shape_params = config.MODEL.NUM_SMPL_BETAS
        self.num_glob_params = 6
        self.num_cam_params = 3
        self.register_buffer("init_glob", rotmat_to_rot6d(torch.eye(3)[None, :].float()))
        self.register_buffer("init_cam", torch.tensor([0.9, 0.0, 0.0]).float())

        if config.MODEL.NUM_RESNET_LAYERS != 18:
            raise NotImplementedError("only the ResNet-18 encoder of the released model is implemented")
        self.image_encoder = resnet18(in_channels=config.MODEL.NUM_IN_CHANNELS, pretrained=False)
        num_image_features, fc1_dim = 512, 512
        embed_dim = config.MODEL.EMBED_DIM

        self.activation = nn.ELU()
        self.fc1 = nn.Linear(num_image_features, fc1_dim)
        self.fc_shape = nn.Linear(fc1_dim, self.num_shape_params * 2)
        self.fc_glob = nn.Linear(fc1_dim, self.num_glob_params)
        self.fc_cam = nn.Linear(fc1_dim, self.num_cam_params)
        self.fc_embed = nn.Linear(num_image_features + self.num_shape_params * 2 + self.num_glob_params
                                  + self.num_cam_params, embed_dim)
        self.fc_pose = nn.ModuleList()
        for joint in range(self.num_joints):
            in_dim = embed_dim + len(self.parents_dict[joint]) * (9 + 3 + 9)
            self.fc_pose.append(nn.Sequential(nn.Linear(in_dim, embed_dim // 2), self.activation,
                                              nn.Linear(embed_dim // 2, 9)))
        # kinematic depth levels: joints whose ancestors are all in earlier levels
        depth = [len(self.parents_dict[j]) for j in range(self.num_joints)]
        self.levels = [[j for j in range(self.num_joints) if depth[j] == d] for d in range(max(depth) + 1)]
        self._prepared = None
        self._pinned_bufs = {}
        self.register_load_state_dict_post_hook(_invalidate_after_load)
        self.composite_head = True     # joint loop through hps_head_pose_levels (one call) instead of per-level Python
        self.svd_mode = "device"       # "device": in-kernel gesdd-faithful SVD; "host": MKL sgesdd round trip (the routine itself)
        self.svd_flavor = None         # None: the rounding flavour of this host's MKL (calibrated); 0 / 1 force one
        self.latency_mode = False      # set_latency_mode(): encoder on direct kernels with many K slices, joint MLPs on wide workgroups

    def set_latency_mode(self, on=True):
        """One switch for one-image-at-a-time deployments (the reference's run_predict operating point): the encoder's latency mode
        (ResNet.set_latency_mode) and 1024-thread / eight-K-slice workgroups for the joint MLPs of the head
        (HPS_HEAD_WIDE_WORKGROUPS; device SVD mode only).  A property of the model: within a mode results do not depend on the
        batch size; between the modes they differ in the last bits (other summation orders)."""
        self.latency_mode = bool(on)
        self.image_encoder.set_latency_mode(on)

    def _flavor(self):
        """Rounding flavour of the in-kernel SVD: ``svd_flavor`` if set (0 reference BLAS rounding, 1 fused), else the one that
        reproduces this host's LAPACK bit for bit (_capi.svd_flavor)."""
        return _capi.svd_flavor() if self.svd_flavor is None else int(self.svd_flavor)

    # ---- kernel-side weights; rebuilt after .to() / load_state_dict ----
    def _apply(self, fn, *args, **kwargs):
        self._prepared = None
        return super()._apply(fn, *args, **kwargs)

    def invalidate(self):
        """Drop the kernel-side weight copies / pointer tables (rebuilt by the next forward).  Automatic after .to() and after
        any load_state_dict that reaches this module, directly or through a parent (post hook); call it by hand after
        editing parameters in place."""
        self._prepared = None

    def __getstate__(self):
        # copy.deepcopy / pickle: the pointer tables hold raw device addresses of THIS module's tensors and the staging
        # buffers are page-locked host memory -- a copy must rebuild its own
        state = self.__dict__.copy()
        state["_prepared"] = None
        state["_pinned_bufs"] = {}
        return state

    def _pinned(self, name, numel):
        """Reusable page-locked host staging buffer (fp32) of at least ``numel`` elements, one per (name, stream): the last
        level's upload is still in flight when forward returns, and only a later forward ON THE SAME STREAM is ordered
        behind it (its first stream synchronisation retires the copy) -- a forward on another stream gets its own buffer."""
        stream = torch.cuda.current_stream()
        key = (name, stream.cuda_stream)
        buf = self._pinned_bufs.get(key)
        if buf is None or buf.numel() < numel:
            if buf is not None:
                stream.synchronize()          # never free a staging block with a copy in flight
            buf = torch.empty(max(numel, 1024), dtype=torch.float32, pin_memory=True)
            self._pinned_bufs[key] = buf
        return buf[:numel]

    def prepare(self):
        dev = self.fc1.weight.device
        t = lambda w: w.detach().float().t().contiguous()
        c = lambda w: w.detach().float().contiguous()
        nsh, ng, nc = self.num_shape_params * 2, self.num_glob_params, self.num_cam_params
        p = {}
        p["fc1_wt"], p["fc1_b"] = t(self.fc1.weight), c(self.fc1.bias)
        # fc_shape | fc_glob | fc_cam fused into one (512 -> 29) layer; init_glob / init_cam as addend
        p["sgc_wt"] = t(torch.cat([self.fc_shape.weight, self.fc_glob.weight, self.fc_cam.weight], dim=0))
        p["sgc_b"] = c(torch.cat([self.fc_shape.bias, self.fc_glob.bias, self.fc_cam.bias]))
        p["sgc_add"] = c(torch.cat([torch.zeros(nsh, device=dev), self.init_glob.reshape(-1), self.init_cam.reshape(-1)]))
        p["embed_wt"], p["embed_b"] = t(self.fc_embed.weight), c(self.fc_embed.bias)
        w1t = [t(m[0].weight) for m in self.fc_pose]
        b1 = [c(m[0].bias) for m in self.fc_pose]
        w2 = [c(m[2].weight) for m in self.fc_pose]
        b2 = [c(m[2].bias) for m in self.fc_pose]
        p["keep"] = (w1t, b1, w2, b2)                      # owners of the device memory behind the pointer tables
        ptrs = lambda ts: torch.tensor([x.data_ptr() for x in ts], dtype=torch.int64, device=dev)
        p["w1t_ptrs"], p["b1_ptrs"], p["w2_ptrs"], p["b2_ptrs"] = ptrs(w1t), ptrs(b1), ptrs(w2), ptrs(b2)
        anc_ptr, anc_idx = [0], []
        for j in range(self.num_joints):
            anc_idx.extend(self.parents_dict[j])
            anc_ptr.append(len(anc_idx))
        p["anc_ptr"] = torch.tensor(anc_ptr, dtype=torch.int32, device=dev)
        p["anc_idx"] = torch.tensor(anc_idx if anc_idx else [0], dtype=torch.int32, device=dev)
        p["levels"] = [torch.tensor(l, dtype=torch.int32, device=dev) for l in self.levels]
        p["level_joints"] = torch.tensor([j for l in self.levels for j in l], dtype=torch.int32, device=dev)
        p["level_sizes_host"] = torch.tensor([len(l) for l in self.levels], dtype=torch.int32)
        p["max_level_size"] = max(len(l) for l in self.levels)
        self._prepared = p
        return p

    # ------------------------------------------------------------------------------------------
    def _trunk(self, feats, p):
        """:95-110: fc1 / ELU, the Gaussian over the betas, glob, cam, the embedding -- three launches (hps_head_trunk); the Gaussian's
        mean / exp(log std), glob and cam come out contiguous: no concatenation buffer, no torch.exp, no clones on the head's stream."""
        B, dev = feats.shape[0], feats.device
        nsh, ng, nc = self.num_shape_params * 2, self.num_glob_params, self.num_cam_params
        embed_dim = self.config.MODEL.EMBED_DIM
        P, s = _capi.ptr, _capi.stream()
        f32 = dict(device=dev, dtype=torch.float32)
        nf = feats.shape[1]
        hidden = p["fc1_wt"].shape[1]
        x = torch.empty(B, hidden, **f32)
        sgc = torch.empty(B, nsh + ng + nc, **f32)
        embed = torch.empty(B, embed_dim, **f32)
        shape_mean = torch.empty(B, self.num_shape_params, **f32)
        shape_scale = torch.empty(B, self.num_shape_params, **f32)
        glob = torch.empty(B, ng, **f32)
        cam = torch.empty(B, nc, **f32)
        _capi.call("hps_head_trunk", P(feats), nf, P(p["fc1_wt"]), P(p["fc1_b"]), P(p["sgc_wt"]), P(p["sgc_b"]), P(p["sgc_add"]),
                   P(p["embed_wt"]), P(p["embed_b"]), P(x), P(sgc), P(embed), P(shape_mean), P(shape_scale), P(glob), P(cam), B, nf,
                   hidden, self.num_shape_params, ng, nc, embed_dim, s)
        return embed, Normal(loc=shape_mean, scale=shape_scale, validate_args=False), glob, cam

    def _pose_buffers(self, B, dev):
        """pose_F, pose_U, pose_S, pose_V, U_proper, S_proper, mode: every joint is in exactly one level and ancestors come from earlier
        levels, so all entries are written before they are read -- no zero fill (seven launches less on the head's stream)."""
        f32 = dict(device=dev, dtype=torch.float32)
        nj = self.num_joints
        return (torch.empty(B, nj, 3, 3, **f32), torch.empty(B, nj, 3, 3, **f32), torch.empty(B, nj, 3, **f32),
                torch.empty(B, nj, 3, 3, **f32), torch.empty(B, nj, 3, 3, **f32), torch.empty(B, nj, 3, **f32),
                torch.empty(B, nj, 3, 3, **f32))

    def forward(self, input, input_feats=None):
        """models/poseMF_shapeGaussian_net.py:85-162.  input: (B,C,D,D); ``input_feats`` skips the encoder."""
        if input_feats is None:
            input_feats = self.image_encoder(input)
        _capi.require_device(input_feats, "input_feats")
        p = self._prepared or self.prepare()
        feats = _capi.f32c(input_feats)
        B = feats.shape[0]
        dev = feats.device
        nj = self.num_joints
        embed_dim = self.config.MODEL.EMBED_DIM
        P, s = _capi.ptr, _capi.stream()
        f32 = dict(device=dev, dtype=torch.float32)
        embed, shape_dist, glob, cam = self._trunk(feats, p)

        # hierarchical pose prediction (:121-160), one kinematic level at a time
        pose_F, pose_U, pose_S, pose_V, U_proper, S_proper, mode = self._pose_buffers(B, dev)
        delta = float(self.config.MODEL.DELTA_I_WEIGHT) if self.config.MODEL.DELTA_I else 0.0
        stream = torch.cuda.current_stream()
        if self.svd_mode not in ("device", "host"):
            raise ValueError("svd_mode must be 'device' or 'host'")
        device_svd = self.svd_mode == "device"
        if self.composite_head:
            # the whole joint loop in one call across the C ABI (csrc/composite.hip: same launches, same order)
            sizes = p["level_sizes_host"]
            max_n = p["max_level_size"]
            VP = _capi._P
            if device_svd:
                f_dev = usv_dev = None
                fh = uh = None
            else:
                n_f = _capi.query_workspace(_capi.WS_HEAD_F, B, max_n) // 4
                n_usv = _capi.query_workspace(_capi.WS_HEAD_USV, B, max_n) // 4
                f_dev = torch.empty(n_f, **f32)
                usv_dev = torch.empty(n_usv, **f32)
                fh, uh = VP(self._pinned("f", n_f).data_ptr()), VP(self._pinned("usv", n_usv).data_ptr())
            _capi.call("hps_head_pose_levels", P(embed), embed_dim, embed_dim // 2, _capi.iptr(p["level_joints"]),
                       VP(sizes.data_ptr()), len(p["levels"]), _capi.iptr(p["anc_ptr"]), _capi.iptr(p["anc_idx"]),
                       VP(p["w1t_ptrs"].data_ptr()), VP(p["b1_ptrs"].data_ptr()), VP(p["w2_ptrs"].data_ptr()),
                       VP(p["b2_ptrs"].data_ptr()), P(U_proper), P(S_proper), P(mode), delta, P(pose_F), P(pose_U), P(pose_S),
                       P(pose_V), P(f_dev), P(usv_dev), fh, uh, B, nj, _SVD_THREADS,
                       ((_capi.SVD_DEVICE_FMA if self._flavor() == _capi.SVD_ROUNDING_FMA else _capi.SVD_DEVICE) |
                        (_capi.HEAD_WIDE_WORKGROUPS if self.latency_mode else 0)) if device_svd else _capi.SVD_HOST, s)
            return pose_F, pose_U, pose_S, pose_V, mode, shape_dist, glob, cam
        for lvl in p["levels"]:
            n_level = lvl.numel()
            if device_svd:
                _capi.call("hps_head_joint_level_svd", P(embed), embed_dim, embed_dim // 2, _capi.iptr(lvl), n_level,
                           _capi.iptr(p["anc_ptr"]), _capi.iptr(p["anc_idx"]),
                           _capi._P(p["w1t_ptrs"].data_ptr()), _capi._P(p["b1_ptrs"].data_ptr()),
                           _capi._P(p["w2_ptrs"].data_ptr()), _capi._P(p["b2_ptrs"].data_ptr()),
                           P(U_proper), P(S_proper), P(mode), delta, P(pose_F), P(pose_U), P(pose_S), P(pose_V), B, nj,
                           self._flavor() | (_capi.HEAD_WIDE_WORKGROUPS if self.latency_mode else 0), s)
                continue
            f_level = torch.empty(B, n_level, 3, 3, **f32)
            _capi.call("hps_head_joint_level", P(embed), embed_dim, embed_dim // 2, _capi.iptr(lvl), n_level,
                       _capi.iptr(p["anc_ptr"]), _capi.iptr(p["anc_idx"]),
                       _capi._P(p["w1t_ptrs"].data_ptr()), _capi._P(p["b1_ptrs"].data_ptr()),
                       _capi._P(p["w2_ptrs"].data_ptr()), _capi._P(p["b2_ptrs"].data_ptr()),
                       P(U_proper), P(S_proper), P(mode), delta, P(pose_F), P(f_level), B, nj, s)
            # host LAPACK SVD of the level's B * n_level 3x3 matrices (:137), see module docstring.
            # Pinned staging buffers; the stream synchronisation also retires the previous level's upload.
            f_host = self._pinned("f", B * n_level * 9).view(B * n_level, 3, 3)
            f_host.copy_(f_level.view(B * n_level, 3, 3), non_blocking=True)
            stream.synchronize()
            usv_host = self._pinned("usv", B * n_level * 21).view(B * n_level, 21)
            _host_svd_packed(f_host, usv_host)
            usv = torch.empty(B, n_level, 21, **f32)
            usv.view(B * n_level, 21).copy_(usv_host, non_blocking=True)
            _capi.call("hps_head_svd_finish", P(usv), _capi.iptr(lvl), n_level, P(pose_U), P(pose_S), P(pose_V),
                       P(U_proper), P(S_proper), P(mode), B, nj, s)
        return pose_F, pose_U, pose_S, pose_V, mode, shape_dist, glob, cam
